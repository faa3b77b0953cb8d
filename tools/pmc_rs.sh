#!/bin/bash
# PMC passes for the resampler kernels (GPU box).  usage: tools/pmc_rs.sh <outdir>
O=${1:-gpurun_out/pmc_rs}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export PMC_FILTER="resample"
$R/tools/pmc.sh $R/$O/sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- python $R/tools/rsbench.py 256
$R/tools/pmc.sh $R/$O/mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE -- python $R/tools/rsbench.py 256
