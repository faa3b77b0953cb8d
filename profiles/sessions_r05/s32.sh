#!/bin/bash
# round 5, session 32: rocprofv3 evidence for the n_fft 512 / 1024 wave kernels with the 512-byte-run stores (final binary):
# --kernel-trace --stats, then FETCH_SIZE and WRITE_SIZE in their own passes
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r5s32; mkdir -p $O
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for cfg in "512 16000" "1024 44100"; do set -- $cfg
  CMD="python $R/tools/kbench.py --nfft $1 --sr $2 --what stft,stftmel,istft --iters 30"
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_$1 -o k -- $CMD > $O/stats_$1.log 2>&1
  timeout 300 rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d $O/fetch_$1 -o k -- $CMD > $O/fetch_$1.log 2>&1
  timeout 300 rocprofv3 --pmc WRITE_SIZE --kernel-trace --output-format csv -d $O/write_$1 -o k -- $CMD > $O/write_$1.log 2>&1
done
python3 - $O $R <<'PY'
import csv, glob, sys, collections, json, hashlib
O, R = sys.argv[1], sys.argv[2]
out = {"lib_sha256": hashlib.sha256(open(R + "/audiotools_amd/lib/libaudiotools_amd.so", "rb").read()).hexdigest(),
       "command": "tools/kbench.py --nfft N --sr SR --what stft,stftmel,istft --iters 30 (B = 512 x 2 ch x 10 s; placement pool on: the first 36 launches of a shape are its calibration)"}
for n in ("512", "1024"):
    st = glob.glob(f"{O}/stats_{n}/**/*kernel_stats.csv", recursive=True)
    ent = {}
    if st:
        rows = list(csv.DictReader(open(st[0])))
        ent["kernel_stats"] = [{k: r[k] for k in ("Name", "Calls", "AverageNs", "MinNs", "MaxNs")} for r in rows[:6]]
    for name in ("fetch", "write"):
        agg = collections.defaultdict(lambda: collections.defaultdict(list))
        for f in glob.glob(f"{O}/{name}_{n}/**/*counter_collection.csv", recursive=True):
            for r in csv.DictReader(open(f)):
                agg[r["Kernel_Name"][:80]][r["Counter_Name"]].append(float(r["Counter_Value"]))
        ent[name] = {k: {c: sum(v) / len(v) for c, v in d.items()} for k, d in agg.items() if "stft" in k}
    out["n_fft_" + n] = ent
json.dump(out, open(O + "/r05_small_sizes_summary.json", "w"), indent=1)
print(json.dumps(out, indent=1)[:5000])
PY
rm -rf $O/stats_* $O/fetch_* $O/write_*
