#!/bin/bash
# PMC passes for the resampler kernels (GPU box): the fp16-split kernel (default) or the f32 MFMA kernel (KIND=mfma).
# usage: [KIND=f16|mfma] tools/pmc_rs.sh <outdir>      (one rocprofv3 --pmc pass per counter group, kernel trace only)
O=${1:-gpurun_out/pmc_rs}
KIND=${KIND:-f16}
R=$GRAFT_REPO_ROOT
mkdir -p $R/$O
export PMC_FILTER="resample"
RUN="python $R/tools/rsbench.py --batch 256 --iters 3 --rounds 1 --only $KIND"
$R/tools/pmc.sh $R/$O/fetch FETCH_SIZE -- $RUN
$R/tools/pmc.sh $R/$O/write WRITE_SIZE -- $RUN
$R/tools/pmc.sh $R/$O/sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- $RUN
$R/tools/pmc.sh $R/$O/mfma SQ_INSTS_MFMA SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VMEM_RD SQ_INST_LEVEL_VMEM GRBM_GUI_ACTIVE -- $RUN
