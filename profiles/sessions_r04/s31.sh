#!/bin/bash
# round 4, session 31: everything on the last binary -- full GPU suite, smoke, the three bench lines, per-kernel rows,
# counters of cfg5 and of the north-star step
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/s31; mkdir -p $O
( timeout 1500 python -m pytest tests -m gpu -q 2>&1 | tail -12 ) > $O/pytest.log 2>&1
tail -4 $O/pytest.log
timeout 200 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; tail -1 $O/smoke.log
timeout 300 python bench.py --steps 20 --warmup 5 > $O/bench.log 2>&1; tail -1 $O/bench.log | cut -c1-2200
timeout 300 python bench.py --config cfg4 --steps 20 --warmup 5 > $O/bench_cfg4.log 2>&1; tail -1 $O/bench_cfg4.log | cut -c1-1500
timeout 300 python bench.py --config cfg5 --steps 10 --warmup 3 > $O/bench_cfg5.log 2>&1; tail -1 $O/bench_cfg5.log | cut -c1-1500
timeout 200 python tools/kbench.py --what stft,stftmel,lufs,lufs3,istft,copy --iters 20 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/kbench.log
timeout 200 python tools/kbench.py --what stftmel,lufs --iters 50 --batch 64 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
for cfg in "16000 400" "24000 1200" "48000 1920" "16000 512" "22050 1024"; do
  set -- $cfg
  echo "### sr=$1 n_fft=$2" | tee -a $O/kbench.log
  timeout 200 python tools/kbench.py --what stft,$( [ $2 -gt 1100 -o $2 -eq 400 ] && echo genmel || echo stftmel ),istft --iters 10 --batch 256 --sr $1 --nfft $2 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee -a $O/kbench.log
done
timeout 200 python tools/rsbench.py --iters 20 --rounds 2 --only f16,mfma 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/rsbench.log
timeout 300 python tools/cfgbench.py 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/cfgbench.log
timeout 300 python tools/tfmbench.py 2>&1 | grep -v -e amdgpu.ids -e "^$" | tee $O/tfmbench.log
bash tools/profile_round.sh r04_cfg5 --config cfg5 > $O/profile_cfg5.log 2>&1
bash tools/profile_round.sh r04_bench > $O/profile_bench.log 2>&1
