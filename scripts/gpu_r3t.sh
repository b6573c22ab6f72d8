cd $GRAFT_REPO_ROOT
O=gpurun_out/r3t; mkdir -p $O
timeout 600 python -m pytest tests/test_iqn_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 900 python scripts/train_variants.py "nt_all:" "nt_all_ntload:-DMN_PLOAD=1" "nt16only:-DMN_PSTORE=2 -DMN_SCALAR_PLAIN=1" "plain:-DMN_PSTORE=0" > $O/variants.log 2>&1; grep -v amdgpu $O/variants.log
