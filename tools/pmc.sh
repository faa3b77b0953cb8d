#!/bin/bash
# usage: tools/pmc.sh <outdir> <counter list...> -- <cmd...>   (one rocprofv3 --pmc pass)
out=$1; shift
ctrs=()
while [ "$1" != "--" ]; do ctrs+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp
rocprofv3 --pmc "${ctrs[@]}" --kernel-trace --output-format csv -d "$out" -o pmc -- "$@" > "$out.log" 2>&1
python3 - "$out" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
files = glob.glob(out + "/**/*counter_collection.csv", recursive=True)
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for f in files:
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"][:60]
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in agg.items():
    import os, re
    if re.search(os.environ.get("PMC_FILTER", "stft|kweight|lufs"), k):
        print(k, {c: round(sum(v) / len(v), 1) for c, v in d.items()}, "n=", len(next(iter(d.values()))))
PY
