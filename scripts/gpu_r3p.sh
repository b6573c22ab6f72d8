cd $GRAFT_REPO_ROOT
O=gpurun_out/r3p; mkdir -p $O
timeout 300 python scripts/rollout_phase_timing.py 4096 0 2>&1 | grep -v amdgpu.ids > $O/rollout_phases.log; cat $O/rollout_phases.log
timeout 900 python -m pytest tests/test_rollout_gpu.py tests/test_env_gpu.py tests/test_step_append_gpu.py -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --no-learner --envs 4096 --steps 1000 --warmup 200 --rollout 100 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rollout 4096', d['value'])"
timeout 300 python bench.py --no-learner --envs 65536 --steps 400 --warmup 50 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step-only 65536', d['value'], d['roofline']['launch_ms'])"
