mkdir -p gpurun_out/s16; cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s16
timeout 200 python tools/cfgbench.py --only lowpass,eq,applyir,chain > $O/cfg4.log 2>&1 < /dev/null; grep -v amdgpu.ids $O/cfg4.log
(cd /tmp && export TMPDIR=/tmp && timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof -o c -- python $GRAFT_REPO_ROOT/tools/cfgbench.py --only lowpass,eq,applyir,chain > $O/prof.log 2>&1 < /dev/null)
f=$(find $O/prof -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/cfg4_kernel_stats.csv; head -32 $O/cfg4_kernel_stats.csv | cut -c1-150; fi
