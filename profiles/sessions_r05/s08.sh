#!/bin/bash
# round 5, session 8: the wave-per-pair row kernel with the N2 = 1000 pass list as literals (37 KB of code instead of 128 KB);
# the convolution tests on the capped plan with the failure of s07 printed in full
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s08; mkdir -p $O
export AT_DEV_KNOBS=1
( AT_LONGCONV_N2MAX=1024 timeout 400 python -m pytest tests/test_gpu_parity.py -m gpu -q -k "conv or apply_ir or longconv" 2>&1 | tail -40 ) > $O/pytest_conv_wave.log 2>&1
cat $O/pytest_conv_wave.log
for rep in 1 2; do
  echo "== shipped plan (N2 = 2000, workgroup per pair)"; AT_ROWCONV_WAVE=0 timeout 120 python tools/convbench.py --iters 10 --engines fourstep 2>&1 | grep -v "^/opt"
  echo "== N2 <= 1024, wave per pair"; AT_LONGCONV_N2MAX=1024 timeout 120 python tools/convbench.py --iters 10 --engines fourstep 2>&1 | grep -v "^/opt"
done > $O/convbench.log 2>&1
cat $O/convbench.log
cd /tmp && export TMPDIR=/tmp
AT_LONGCONV_N2MAX=1024 timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -o conv -- python $GRAFT_REPO_ROOT/tools/convbench.py --iters 10 --engines fourstep > $O/kt.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/kt/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:4]: print(r["Name"][:70], r["Calls"], round(float(r["AverageNs"])/1e3,1))
PY
