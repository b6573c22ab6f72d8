cd $GRAFT_REPO_ROOT
O=gpurun_out/r3c; mkdir -p $O
export TMPDIR=/tmp
timeout 300 python scripts/train_phase_timing.py 256 300 > $O/phases.log 2>&1; cat $O/phases.log
timeout 900 python scripts/train_variants.py "c32s8:" "c16s16:-DMN_RED_COLS=16 -DMN_RED_SEG=16" "c64s4:-DMN_RED_COLS=64 -DMN_RED_SEG=4" "c32s16:-DMN_RED_COLS=32 -DMN_RED_SEG=16" "c8s32:-DMN_RED_COLS=8 -DMN_RED_SEG=32" "c64s8:-DMN_RED_COLS=64 -DMN_RED_SEG=8" > $O/variants.log 2>&1; cat $O/variants.log
timeout 600 python -m pytest tests/test_iqn_gpu.py tests/test_multigpu_paths_gpu.py -x -q -m gpu -k "not config3_size" 2>&1 | tail -30 > $O/pytest_iqn.log; cat $O/pytest_iqn.log
