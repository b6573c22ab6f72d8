cd $GRAFT_REPO_ROOT
O=gpurun_out/r3h; mkdir -p $O
timeout 900 python scripts/pingpong_probe.py > $O/pingpong_probe.log 2>&1; cat $O/pingpong_probe.log
