cd $GRAFT_REPO_ROOT
O=gpurun_out/r3e; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_iqn_gpu.py tests/test_multigpu_paths_gpu.py -x -q -m gpu -k "not config3_size" 2>&1 | tail -30 > $O/pytest_iqn.log; cat $O/pytest_iqn.log
timeout 300 python scripts/train_phase_timing.py 256 300 > $O/phases_staged.log 2>&1; cat $O/phases_staged.log
MN_TRAIN_FLAGS=0 timeout 300 python scripts/train_phase_timing.py 256 300 2>&1 | head -8 > $O/phases_unstaged.log; cat $O/phases_unstaged.log
