cd $GRAFT_REPO_ROOT
timeout 900 bash tools/profile_round.sh r02b < /dev/null > gpurun_out/s22_round.log 2>&1
timeout 600 bash tools/profile_rows.sh r02b < /dev/null > gpurun_out/s22_rows.log 2>&1
tail -5 gpurun_out/s22_rows.log
