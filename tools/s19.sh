mkdir -p gpurun_out/s19; cd $GRAFT_REPO_ROOT
L=gpurun_out/s19/log.txt
echo "default 256x3" >> $L; timeout 100 python tools/convbench.py --engines fourstep >> $L 2>&1 < /dev/null
for t in 512 1024; do
echo "threads $t" >> $L; AT_LIB_PATH=audiotools_amd/lib/libaudiotools_amd_t$t.so timeout 100 python tools/convbench.py >> $L 2>&1 < /dev/null
done
grep -v amdgpu.ids $L
