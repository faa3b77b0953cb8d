#!/bin/bash
# round 3, session 51: trace + counters of the last binary's 2048-point kernels (the inverse kernel changed after s88)
cd $GRAFT_REPO_ROOT
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/s91; mkdir -p $O
cd /tmp && export TMPDIR=/tmp
timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $O/k512 -o k -- python $R/tools/kbench.py --what stft,stftmel,lufs,istft --iters 20 --batch 512 > $O/k512.log 2>&1
f=$(find $O/k512 -name "*kernel_stats.csv" | head -1); [ -n "$f" ] && cp "$f" $O/r03_kernels_b512_kernel_stats.csv; rm -rf $O/k512
grep "ms " $O/k512.log
head -6 $O/r03_kernels_b512_kernel_stats.csv | cut -c1-170
cd $R
export PMC_FILTER="istft"
timeout 100 bash tools/pmc.sh $O/sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python $R/tools/kbench.py --what istft --iters 5 | tee $O/sq.txt
timeout 100 bash tools/pmc.sh $O/lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY -- python $R/tools/kbench.py --what istft --iters 5 | tee $O/lds.txt
rm -rf $O/sq $O/lds
