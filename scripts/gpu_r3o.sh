cd $GRAFT_REPO_ROOT
O=gpurun_out/r3o; mkdir -p $O
for a in "4096 0" "4096 4" "4096 2" "65536 0"; do timeout 300 python scripts/rollout_phase_timing.py $a 2>&1 | grep -v amdgpu.ids; done > $O/rollout_phases.log; cat $O/rollout_phases.log
timeout 600 python -m pytest tests/test_rollout_gpu.py tests/test_iqn_gpu.py -x -q -m gpu 2>&1 | tail -3
