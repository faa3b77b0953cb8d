mkdir -p gpurun_out/s20; cd $GRAFT_REPO_ROOT
L=gpurun_out/s20/log.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fourstep or room" > gpurun_out/s20/pytest.log 2>&1 < /dev/null; tail -2 gpurun_out/s20/pytest.log
echo "default 256x3" >> $L; timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
echo "threads 512" >> $L; AT_LIB_PATH=audiotools_amd/lib/libaudiotools_amd_t512.so timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
grep -v amdgpu.ids $L
