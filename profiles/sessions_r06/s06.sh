#!/bin/bash
# round 6, session 6: the abort of s05's pytest selection (test_istft_autograd_native): alone, in the selection, then the whole suite
cd $GRAFT_REPO_ROOT
O=$GRAFT_REPO_ROOT/gpurun_out/r6s06; mkdir -p $O
echo "### alone"; timeout 600 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "istft_autograd_native" 2>&1 | grep -v "^Extension modules" | tail -15
echo "### selection"; timeout 1200 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "mel or stft or north_star or cfg2 or cfg5" 2>&1 | grep -v "^Extension modules" | tail -15
echo "### whole suite"; timeout 2400 python -m pytest tests -x -q -m gpu > $O/pytest.log 2>&1; grep -v "^Extension modules" $O/pytest.log | tail -12
dmesg 2>/dev/null | tail -5
