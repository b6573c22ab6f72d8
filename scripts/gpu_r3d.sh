cd $GRAFT_REPO_ROOT
O=gpurun_out/r3d; mkdir -p $O
export TMPDIR=/tmp
timeout 600 python -m pytest tests/test_iqn_gpu.py tests/test_multigpu_paths_gpu.py -x -q -m gpu -k "not config3_size and not two_ranks" 2>&1 | tail -30 > $O/pytest_iqn.log; cat $O/pytest_iqn.log
timeout 300 python scripts/train_phase_timing.py 256 300 > $O/phases_staged.log 2>&1; cat $O/phases_staged.log
MN_TRAIN_FLAGS=0 timeout 300 python scripts/train_phase_timing.py 256 300 > $O/phases_unstaged.log 2>&1; cat $O/phases_unstaged.log
timeout 900 python scripts/train_variants.py "default:" > $O/variants.log 2>&1; cat $O/variants.log
MN_TRAIN_FLAGS=0 timeout 900 python scripts/train_variants.py "default:" > $O/variants_unstaged.log 2>&1; cat $O/variants_unstaged.log
timeout 300 python scripts/learner_bench.py 3000 > $O/learner_bench.log 2>&1; cat $O/learner_bench.log
cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/$O/prof_learner -- python $GRAFT_REPO_ROOT/scripts/learner_bench.py 500 > /dev/null 2>&1
cd $GRAFT_REPO_ROOT
find $O/prof_learner -name "*kernel_stats.csv" | head -1 | xargs -I{} head -5 {} | cut -c1-160 > $O/learner_kernel_stats.txt; cat $O/learner_kernel_stats.txt
timeout 600 python -m pytest tests/test_multigpu_paths_gpu.py -x -q -m gpu -k "two_ranks" 2>&1 | tail -40 > $O/pytest_two_ranks.log; cat $O/pytest_two_ranks.log
