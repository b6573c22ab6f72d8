cd $GRAFT_REPO_ROOT
timeout 900 python -m pytest tests/test_env_gpu.py -q -m gpu -s -k "g3_single_step_golden or mixed_single_step or strict" 2>&1 | grep -E "observed|passed|failed"
