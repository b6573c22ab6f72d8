cd $GRAFT_REPO_ROOT
O=gpurun_out/r3g; mkdir -p $O
timeout 600 python scripts/overlap_probe.py > $O/overlap_probe.log 2>&1; cat $O/overlap_probe.log
