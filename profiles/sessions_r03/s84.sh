#!/bin/bash
# round 3, session 44: where the 512-point (16 kHz) forward kernel's time goes -- counters of kbench at n_fft 512
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s84; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/tools/kbench.py --what stft,stftmel --iters 5 --sr 16000 --nfft 512"
export PMC_FILTER="stft"
timeout 200 python tools/kbench.py --what stft,stftmel,istft --iters 20 --sr 16000 --nfft 512 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/k.log
timeout 200 bash tools/pmc.sh $O/fetch FETCH_SIZE -- $CMD | tee $O/fetch.txt
timeout 200 bash tools/pmc.sh $O/write WRITE_SIZE -- $CMD | tee $O/write.txt
timeout 200 bash tools/pmc.sh $O/sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- $CMD | tee $O/sq.txt
timeout 200 bash tools/pmc.sh $O/lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -- $CMD | tee $O/lds.txt
timeout 200 bash tools/pmc.sh $O/mem SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY -- $CMD | tee $O/mem.txt
rm -rf $O/fetch $O/write $O/sq $O/lds $O/mem
