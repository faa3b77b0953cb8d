#!/bin/bash
# rocprofv3 kernel stats (+ optional PMC) of the four-step convolution bench (GPU box).  usage: tools/profile_conv.sh <tag> [pmc]
tag=${1:-r02}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/conv_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k -o c -- python $R/tools/convbench.py --iters 5 > $O/conv.log 2>&1 < /dev/null
f=$(find $O/k -name "*kernel_stats.csv" | head -1)
if [ -n "$f" ]; then cp "$f" $O/kernel_stats.csv; head -12 $O/kernel_stats.csv | cut -c1-220; fi
cat $O/conv.log | tail -4
if [ "$2" = "pmc" ]; then
  export PMC_FILTER="colfft|rowconv"
  timeout 200 $R/tools/pmc.sh $O/pmc_fetch FETCH_SIZE -- python $R/tools/convbench.py --iters 2 --engines fourstep < /dev/null
  timeout 200 $R/tools/pmc.sh $O/pmc_write WRITE_SIZE -- python $R/tools/convbench.py --iters 2 --engines fourstep < /dev/null
  timeout 200 $R/tools/pmc.sh $O/pmc_sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/convbench.py --iters 2 --engines fourstep < /dev/null
fi
