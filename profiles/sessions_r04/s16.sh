#!/bin/bash
# round 4, session 16: counters of the speech-window tiles (forward plan 3 / 5, one-pass inverse)
cd $GRAFT_REPO_ROOT
O=gpurun_out/s16; mkdir -p $GRAFT_REPO_ROOT/$O
export PMC_FILTER="generic"
for cfg in "16000 400" "48000 1920"; do
  set -- $cfg
  RUN="python $GRAFT_REPO_ROOT/tools/kbench.py --what stft,istft --iters 3 --batch 256 --sr $1 --nfft $2"
  echo "### n_fft $2" | tee -a $GRAFT_REPO_ROOT/$O/pmc.txt
  tools/pmc.sh $GRAFT_REPO_ROOT/$O/sq_$2 SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- $RUN | tee -a $GRAFT_REPO_ROOT/$O/pmc.txt
  cd $GRAFT_REPO_ROOT
  tools/pmc.sh $GRAFT_REPO_ROOT/$O/lds_$2 SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SALU SQ_INST_LEVEL_VMEM SQ_WAIT_INST_LDS GRBM_GUI_ACTIVE SQ_INSTS_VMEM_RD -- $RUN | tee -a $GRAFT_REPO_ROOT/$O/pmc.txt
  cd $GRAFT_REPO_ROOT
  tools/pmc.sh $GRAFT_REPO_ROOT/$O/fetch_$2 FETCH_SIZE -- $RUN | tee -a $GRAFT_REPO_ROOT/$O/pmc.txt
  cd $GRAFT_REPO_ROOT
  tools/pmc.sh $GRAFT_REPO_ROOT/$O/write_$2 WRITE_SIZE -- $RUN | tee -a $GRAFT_REPO_ROOT/$O/pmc.txt
  cd $GRAFT_REPO_ROOT
done
rm -rf $GRAFT_REPO_ROOT/$O/sq_* $GRAFT_REPO_ROOT/$O/lds_*[0-9] $GRAFT_REPO_ROOT/$O/fetch_*[0-9] $GRAFT_REPO_ROOT/$O/write_*[0-9] 2>/dev/null
