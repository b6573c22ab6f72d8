set -x
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r3a
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_iqn_gpu.py tests/test_multigpu_paths_gpu.py -x -q -m gpu -k "not config3_size" 2>&1 | tail -15 > gpurun_out/r3a/pytest_iqn.log
cat gpurun_out/r3a/pytest_iqn.log
timeout 300 python scripts/learner_bench.py 3000 > gpurun_out/r3a/learner_bench.log 2>&1
cat gpurun_out/r3a/learner_bench.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r3a/prof_learner -- python $GRAFT_REPO_ROOT/scripts/learner_bench.py 500 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find gpurun_out/r3a/prof_learner -name "*kernel_stats.csv" | head -1 | xargs -I{} head -12 {} > gpurun_out/r3a/learner_kernel_stats.txt
cat gpurun_out/r3a/learner_kernel_stats.txt
timeout 900 python bench.py > gpurun_out/r3a/bench.json 2> gpurun_out/r3a/bench.err
tail -3 gpurun_out/r3a/bench.err
cat gpurun_out/r3a/bench.json
