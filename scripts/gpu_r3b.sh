set -x
cd $GRAFT_REPO_ROOT
O=gpurun_out/r3b; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/train_phase_timing.py 256 300 > $O/phases.log 2>&1; cat $O/phases.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_learner -- python $GRAFT_REPO_ROOT/scripts/learner_bench.py 500 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_learner -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {} > $O/learner_kernel_stats.txt; cat $O/learner_kernel_stats.txt
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
