mkdir -p gpurun_out/s21; cd $GRAFT_REPO_ROOT
timeout 700 python -m pytest tests -m gpu -q > gpurun_out/s21/pytest.log 2>&1 < /dev/null; tail -4 gpurun_out/s21/pytest.log
timeout 500 bash tools/rows.sh > gpurun_out/s21/rows.txt 2>&1 < /dev/null
timeout 200 python bench.py --steps 20 --warmup 3 > gpurun_out/s21/bench_ns.log 2>&1 < /dev/null
timeout 200 python bench.py --config cfg4 --steps 10 > gpurun_out/s21/bench_cfg4.log 2>&1 < /dev/null
timeout 200 python bench.py --config cfg5 --batch 256 --steps 10 > gpurun_out/s21/bench_cfg5.log 2>&1 < /dev/null
timeout 100 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/s21/smoke.log 2>&1 < /dev/null; tail -1 gpurun_out/s21/smoke.log
