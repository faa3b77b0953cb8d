#!/bin/bash
cd $GRAFT_REPO_ROOT
O=gpurun_out/s79; mkdir -p $O
timeout 200 python -m pytest tests/test_gpu_parity.py -m gpu -q -x -k "alter_drr_reports" 2>&1 | grep -v "^$" | tail -40 | tee $O/t.log
