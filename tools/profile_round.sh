#!/bin/bash
# Collect the round's profile evidence on the GPU box; summaries land in gpurun_out/profile_<tag>/
# (copy what you want judged into profiles/).   usage: tools/profile_round.sh <tag> [extra bench args]
tag=${1:-r02}; shift
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/profile_$tag
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
CMD="python $R/bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-share $*"
# 1. kernel trace + stats for the bench command
# (many more steps than the counter passes: the per-kernel AVERAGE of this file is what bench.py's roofline.avg_launch_ms is
#  checked against, and it also holds the launches outside the timed loop -- the first ones after start-up while clocks
#  ramp, the ~36 calibration launches of the output placement pool into candidate buffers that are then dropped, the
#  plain-allocation probe of roofline.placement: 200 timed steps keep them below a quarter of the calls)
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o bench -- python $R/bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-share --no-probes $* > $O/bench_stats.log 2>&1
# 2. HBM traffic counters, each in its own pass (guide: FETCH_SIZE 3 TCC slots, WRITE_SIZE 2)
timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/pmc_fetch -o bench -- $CMD > $O/pmc_fetch.log 2>&1
timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/pmc_write -o bench -- $CMD > $O/pmc_write.log 2>&1
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_INSTS_LDS SQ_ACTIVE_INST_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY --kernel-trace --output-format csv -d $O/pmc_sq -o bench -- $CMD > $O/pmc_sq.log 2>&1
timeout 300 rocprofv3 --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $O/pmc_lds -o bench -- $CMD > $O/pmc_lds.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_INST_LEVEL_VMEM SQ_INSTS_SALU SQ_ACTIVE_INST_SCA SQ_INSTS_MFMA --kernel-trace --output-format csv -d $O/pmc_mem -o bench -- $CMD > $O/pmc_mem.log 2>&1
python3 - $O <<'PY'
import csv, glob, sys, collections, json, os, socket
O = sys.argv[1]
import subprocess
try:
    uid = [l.split(":")[-1].strip() for l in subprocess.run(["rocm-smi", "--showuniqueid"], capture_output=True, text=True, timeout=30).stdout.splitlines() if "Unique ID" in l and "GPU[" in l][0]
except Exception:
    uid = "unknown"
import hashlib
R = os.environ.get("GRAFT_REPO_ROOT", ".")
lib = os.environ.get("AT_LIB_PATH") or os.path.join(R, "audiotools_amd", "lib", "libaudiotools_amd.so")
out = {"lib_sha256": hashlib.sha256(open(lib, "rb").read()).hexdigest(), "box": "MI355X unique id " + uid, "stats_command": "bench.py --steps 200 --warmup 10 --no-cpu-baseline --no-share --no-probes", "command": "bench.py --steps 10 --warmup 3 --no-cpu-baseline --no-share (one rocprofv3 pass per counter group)"}
st = glob.glob(O + "/stats/**/*kernel_stats.csv", recursive=True)
if st:
    rows = list(csv.DictReader(open(st[0])))
    out["kernel_stats"] = [{k: r[k] for k in r if k in ("Name", "Calls", "TotalDurationNs", "AverageNs", "Percentage", "MinNs", "MaxNs")} for r in rows[:14]]
    import shutil
    shutil.copy(st[0], O + "/kernel_stats.csv")
for name in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_lds", "pmc_mem"):
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(O + "/" + name + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            agg[r["Kernel_Name"][:70]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    out[name] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if "elementwise" not in k and "distribution" not in k}
json.dump(out, open(O + "/summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:7000])
PY
