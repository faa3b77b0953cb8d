#!/bin/bash
# round 3, session 6: PMC survey of the main kernels (LDS conflicts / wait split) + filterbank test
cd $GRAFT_REPO_ROOT
O=gpurun_out/s45; mkdir -p $O
R=$GRAFT_REPO_ROOT
( timeout 200 python -m pytest tests -m gpu -q -x -k "filterbank or equalizer" 2>&1 | tail -5 ) > $O/pytest.log 2>&1
tail -3 $O/pytest.log
export PMC_FILTER="stft_mel_kernel_v2|istft_fused|kweight|fir_fft|resample_mfma|lufs_gate"
bash tools/pmc.sh $R/$O/p1 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/kbench.py --what stftmel,istft,lufs --iters 3 --batch 512 > $R/$O/pmc_main.txt 2>&1
bash tools/pmc.sh $R/$O/p2 SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/cfgbench.py --only lowpass,cfg5 > $R/$O/pmc_cfg.txt 2>&1
rm -rf $R/$O/p1 $R/$O/p2
cd $R; cat $O/pmc_main.txt $O/pmc_cfg.txt | grep -v "amdgpu\|^E2026\|^W2026" | cut -c1-600
