mkdir -p gpurun_out/s14; cd $GRAFT_REPO_ROOT
L=gpurun_out/s14/log.txt
for n in 2048 1600 1000 800 500 400; do echo "N2MAX=$n" >> $L; AT_LONGCONV_N2MAX=$n timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null; done
echo "lb3" >> $L; AT_LIB_PATH=audiotools_amd/lib/libaudiotools_amd_lb3.so timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
echo "lb3 N2MAX=1000" >> $L; AT_LONGCONV_N2MAX=1000 AT_LIB_PATH=audiotools_amd/lib/libaudiotools_amd_lb3.so timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
echo "default kbench" >> $L; timeout 150 python tools/kbench.py --what istft,lufs --iters 20 >> $L 2>&1 < /dev/null
echo "noslp kbench" >> $L; AT_LIB_PATH=audiotools_amd/lib/libaudiotools_amd_noslp.so timeout 150 python tools/kbench.py --what istft,lufs --iters 20 >> $L 2>&1 < /dev/null
echo "default firbench" >> $L; timeout 100 python tools/firbench.py 677 >> $L 2>&1 < /dev/null
echo "noslp firbench" >> $L; AT_LIB_PATH=audiotools_amd/lib/libaudiotools_amd_noslp.so timeout 100 python tools/firbench.py 677 >> $L 2>&1 < /dev/null
grep -v amdgpu.ids $L
