cd $GRAFT_REPO_ROOT
O=gpurun_out/r3k; mkdir -p $O
timeout 600 python -m pytest tests/test_split_loop_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/pytest_split.log; cat $O/pytest_split.log
timeout 900 python scripts/split_probe.py > $O/split_probe.log 2>&1; cat $O/split_probe.log
