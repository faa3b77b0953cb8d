mkdir -p gpurun_out/s15; cd $GRAFT_REPO_ROOT
L=gpurun_out/s15/log.txt
timeout 300 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "fourstep or convolve or apply_ir or fir_fft or istft" > gpurun_out/s15/pytest.log 2>&1 < /dev/null; tail -3 gpurun_out/s15/pytest.log
echo "default" >> $L; timeout 100 python tools/convbench.py >> $L 2>&1 < /dev/null
echo "lb3" >> $L; AT_LIB_PATH=audiotools_amd/lib/libaudiotools_amd_lb3.so timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
echo "N2MAX 1600" >> $L; AT_LONGCONV_N2MAX=1600 timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
echo "N2MAX 1000" >> $L; AT_LONGCONV_N2MAX=1000 timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
grep -v amdgpu.ids $L
