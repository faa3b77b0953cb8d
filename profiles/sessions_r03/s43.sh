#!/bin/bash
# round 3, session 4: tiled generic STFT v2 (static vm order, LDS tables, chunked mel) + PMC of rowconv / tiled kernel
cd $GRAFT_REPO_ROOT
O=gpurun_out/s43; mkdir -p $O
R=$GRAFT_REPO_ROOT
( timeout 300 python -m pytest tests -m gpu -q -x -k "generic" 2>&1 | tail -8 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
( timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 256 --sr 96000 --nfft 4096
  timeout 120 python tools/kbench.py --what stft,genmel --iters 10 --batch 128 --sr 192000 --nfft 8192
  timeout 120 python tools/kbench.py --what stft --iters 10 --batch 256 --sr 48000 --nfft 1920 ) > $O/generic.log 2>&1
grep -v amdgpu $O/generic.log
export PMC_FILTER="rowconv|tiled|colfft"
bash tools/pmc.sh $R/$O/pmc_rowconv SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/cfgbench.py --only applyir > $R/$O/pmc_rowconv.txt 2>&1
bash tools/pmc.sh $R/$O/pmc_rowconv2 SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_WAIT_INST_LDS SQ_INST_CYCLES_SALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE -- python $R/tools/cfgbench.py --only applyir > $R/$O/pmc_rowconv2.txt 2>&1
bash tools/pmc.sh $R/$O/pmc_gen SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -- python $R/tools/kbench.py --what stft --iters 3 --batch 256 --sr 96000 --nfft 4096 > $R/$O/pmc_gen.txt 2>&1
rm -rf $R/$O/pmc_rowconv $R/$O/pmc_rowconv2 $R/$O/pmc_gen
cd $R; cat $O/pmc_rowconv.txt $O/pmc_rowconv2.txt $O/pmc_gen.txt | grep -v "amdgpu\|^E2026\|^W2026" | cut -c1-700
