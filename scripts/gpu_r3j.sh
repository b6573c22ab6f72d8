cd $GRAFT_REPO_ROOT
O=gpurun_out/r3j; mkdir -p $O
timeout 900 python scripts/split_probe.py > $O/split_probe.log 2>&1; cat $O/split_probe.log
