#!/bin/bash
# round 4, session 24: the row kernel of the four-step convolution with shorter rows (N2 <= 1024): A) 128-thread workgroups,
# six per CU; B) 256 threads, 123 registers, four per CU; C) the shipped build with AT_LONGCONV_N2MAX=1000
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s24; mkdir -p $O
run() { echo "### $1" | tee -a $O/convbench.log; shift; env "$@" timeout 200 python tools/convbench.py --engines fourstep --iters 10 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/convbench.log; }
for rep in 1 2; do
run "shipped (256 threads, N2 = 2000)" AT_X=0
run "shipped build, AT_LONGCONV_N2MAX=1000" AT_LONGCONV_N2MAX=1000
run "variant A (128 threads x 6, N2 = 1000)" AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_a/libat.so
run "variant B (256 threads x 4, N2 = 1000)" AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_b/libat.so
done
AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_b/libat.so timeout 300 python -m pytest tests -m gpu -q -x -k "conv or apply_ir or ir_" 2>&1 | tail -3 | tee $O/pytest_var_b.log
cd /tmp && export TMPDIR=/tmp
AT_LIB_PATH=$GRAFT_REPO_ROOT/audiotools_amd/lib/var_b/libat.so timeout 200 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_b -o b -- python $GRAFT_REPO_ROOT/tools/convbench.py --engines fourstep --iters 10 > $O/prof_b.log 2>&1
python - <<'PY'
import csv, glob, os
for f in glob.glob(os.path.join(os.environ["GRAFT_REPO_ROOT"], "gpurun_out/s24/prof_b/**/*kernel_stats.csv"), recursive=True):
    for r in list(csv.DictReader(open(f)))[:6]:
        print(r["Name"][:80], r["Calls"], r["AverageNs"], r["Percentage"])
PY
