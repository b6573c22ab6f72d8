cd $GRAFT_REPO_ROOT
O=gpurun_out/r3x; mkdir -p $O
timeout 1200 python -m pytest tests/test_rollout_gpu.py tests/test_env_gpu.py tests/test_step_append_gpu.py -x -q -m gpu > $O/t.log 2>&1; grep -E "passed|failed" $O/t.log
for a in "4096 0" "65536 0"; do timeout 300 python scripts/rollout_phase_timing.py $a 2>&1 | grep -v amdgpu.ids; done > $O/rollout_phases.log; head -12 $O/rollout_phases.log | cut -c1-150
for n in 4096 65536; do
timeout 300 python bench.py --no-learner --envs $n --steps 500 --warmup 100 --rollout 100 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rollout $n T=100', round(d['value']/1e6,1), d['roofline']['frac'])"
done
timeout 300 python bench.py --no-learner --envs 4096 --steps 1000 --warmup 250 --rollout 250 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('rollout 4096 T=250', round(d['value']/1e6,1))"
timeout 300 python bench.py --no-learner --envs 65536 --steps 200 --warmup 50 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('step 65536', round(d['value']/1e6,1), d['ms_per_step'])"
