cd $GRAFT_REPO_ROOT
echo "A = baseline (HEAD), B = v_fma_mixlo/hi_f16 split"
bash scripts/ab_libs.sh
timeout 900 python -m pytest tests/test_act_split_gpu.py tests/test_act_ctx_gpu.py tests/test_iqn_gpu.py -x -q -m gpu 2>&1 | tail -3
