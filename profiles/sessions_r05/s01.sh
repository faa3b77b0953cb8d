#!/bin/bash
# round 5, session 1: the twin's two regimes -- placement (independent buffer sets in one process), schedule knobs, clocks,
# translation / EA counters per buffer set.  Development build (AT_DEV_KNOBS=1 -> lib/libaudiotools_amd_dev.so).
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s01; mkdir -p $O
export AT_DEV_KNOBS=1
( rocm-smi --showclocks --showpower --showmeminfo vram 2>&1 | head -40; amd-smi static -g 0 2>&1 | head -60 ) > $O/smi_static.log 2>&1
timeout 200 python tools/regime.py --sets 4 --rawmalloc --sched --smi --tag untraced-1 > $O/regime1.log 2>&1
timeout 100 python tools/regime.py --sets 4 --tag untraced-2 > $O/regime2.log 2>&1
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --output-format csv -d $O/kt -o kt -- python $GRAFT_REPO_ROOT/tools/regime.py --sets 3 --tag traced > $O/regime_traced.log 2>&1
pass() { n=$1; shift; timeout 120 rocprofv3 --pmc "$@" --kernel-trace --output-format csv -d $O/pmc$n -o pmc -- python $GRAFT_REPO_ROOT/tools/regime.py --pmc --sets 3 --tag pmc$n > $O/pmc$n.log 2>&1; }
pass 1 GRBM_GUI_ACTIVE GRBM_UTCL2_BUSY TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum
pass 2 TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_DRAM_CREDIT_STALL_sum TCC_TOO_MANY_EA_WRREQS_STALL_sum TCC_TAG_STALL_sum
pass 3 TCC_EA0_WRREQ_LEVEL_sum TCC_EA0_WRREQ_sum TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum
pass 4 TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum
cd $GRAFT_REPO_ROOT
find $O -name "*.csv" -size +2000k -delete
tail -n 30 $O/regime1.log
