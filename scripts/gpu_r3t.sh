cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
timeout 600 python -m pytest tests/test_iqn_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python scripts/train_variants.py "base:" "adam_nt:-DMN_ADAM_NT=1" > $O/variants.log 2>&1; grep -v amdgpu $O/variants.log
