#!/bin/bash
# round 5, session 2: which buffer's placement decides the regime (combos), is the slowness uniform inside a buffer (slices),
# loads vs stores (dev twins of the twin), one big pool vs separate allocations, per-channel request counts.
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s02; mkdir -p $O
export AT_DEV_KNOBS=1
timeout 200 python tools/regime.py --sets 5 --combo --slices --variants --tag combos > $O/regime_combo.log 2>&1
timeout 200 python tools/regime.py --sets 1 --pool 5 --slices --tag pool > $O/regime_pool.log 2>&1
cd /tmp && export TMPDIR=/tmp
pass() { n=$1; fmt=$2; shift; shift; timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format $fmt -d $O/pmc$n -o pmc -- python $GRAFT_REPO_ROOT/tools/regime.py --pmc --sets 5 --tag pmc$n > $O/pmc$n.log 2>&1; }
pass 1 csv TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_EA0_RDREQ_DRAM_CREDIT_STALL_sum TCC_BUSY_sum TCC_TAG_STALL_sum
pass 2 csv TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
pass 3 json TCC_EA0_WRREQ TCC_EA0_WRREQ_STALL
pass 4 csv TCC_HIT_sum TCC_MISS_sum TCC_EA0_WRREQ_64B_sum TCC_WRITEBACK_sum
cd $GRAFT_REPO_ROOT
find $O -name "*.json" -size +6000k -delete
find $O -name "*.csv" -size +2000k -delete
tail -n 45 $O/regime_combo.log; tail -n 12 $O/regime_pool.log
