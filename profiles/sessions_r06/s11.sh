#!/bin/bash
# round 6, session 11: counters of the speech-window transforms (n_fft 400 / hop 160 @ 16 kHz, B = 256): where the generic tiled kernels spend their time
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s11; mkdir -p $O
CMD="python $GRAFT_REPO_ROOT/tools/kbench.py --nfft 400 --sr 16000 --batch 256 --what stft,genmel,istft --iters 10 --placed 0"
$CMD 2>&1 | grep -v Warn | grep -v amdgpu.ids
export PMC_FILTER="generic|tiled|ola"
bash tools/pmc.sh $O/sq SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY -- $CMD
bash tools/pmc.sh $O/lds SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD -- $CMD
cd /tmp && export TMPDIR=/tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o k -- $CMD > $O/stats.log 2>&1
python3 - $O <<'PY'
import csv,glob,sys
f=glob.glob(sys.argv[1]+"/stats/**/*kernel_stats.csv",recursive=True)
for r in list(csv.DictReader(open(f[0])))[:8]: print(r["Name"][:80], r["Calls"], float(r["AverageNs"])/1e6)
f=glob.glob(sys.argv[1]+"/stats/**/*kernel_trace.csv",recursive=True)
seen=set()
for r in csv.DictReader(open(f[0])):
    k=r["Kernel_Name"][:60]
    if k in seen: continue
    seen.add(k); print(k, "VGPR", r.get("VGPR_Count"), "LDS", r.get("LDS_Block_Size"), "grid", r.get("Grid_Size"), "wg", r.get("Workgroup_Size"))
PY
