#!/bin/bash
# Per-row measurements of DESIGN.md section 7 in one go (GPU box).  usage: tools/rows.sh > gpurun_out/rows.txt
cd "$(dirname "$0")/.."
run() { echo "### $*"; timeout 300 "$@" 2>&1 < /dev/null | grep -v -e amdgpu.ids -e "^$"; }
run python tools/kbench.py --what stft,stftmel,lufs,istft --iters 20
run python tools/kbench.py --what stftmel,lufs --iters 50 --batch 64
run python tools/kbench.py --what stft,genmel,istft --iters 10 --batch 256 --sr 96000 --nfft 4096
run python tools/kbench.py --what stft,genmel,istft --iters 10 --batch 128 --sr 192000 --nfft 8192
run python tools/cfgbench.py
run python tools/convbench.py
run python tools/firbench.py 153
run python tools/firbench.py 677
run python tools/firbench.py 1047
run python tools/firbench.py 677 direct
run python tools/specbench.py 256
run python tools/gradbench.py 128
run python tools/h2dbench.py 512
