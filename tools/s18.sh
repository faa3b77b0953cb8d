mkdir -p gpurun_out/s18; cd $GRAFT_REPO_ROOT
O=gpurun_out/s18
timeout 400 python -m pytest tests/test_gpu_parity.py tests/test_transforms.py -m gpu -q -x -k "fourstep or room or convolve or apply_ir or cfg4 or impulse or Room or loader_transforms" > $O/pytest.log 2>&1 < /dev/null; tail -4 $O/pytest.log
timeout 200 python tools/cfgbench.py --only lowpass,eq,applyir,chain > $O/cfg4.log 2>&1 < /dev/null; grep -v amdgpu.ids $O/cfg4.log
