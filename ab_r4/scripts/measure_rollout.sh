# configs[1]: where a rollout step goes (ablation build, s_memtime stamps) and the rollout / single-launch rates against the batch size
# (run on the GPU box from the repo root; scripts/measure_round3.sh calls it)
R=${GRAFT_REPO_ROOT:-.}; O=$R/gpurun_out/m3; mkdir -p $O
cd $R
for a in "4096 0" "4096 8" "4096 4" "4096 2" "65536 0"; do python scripts/rollout_phase_timing.py $a 2>&1 | grep -v amdgpu.ids; done > $O/rollout_phases.txt
for n in 4096 16384 65536 262144; do
  python bench.py --no-learner --envs $n --steps 500 --warmup 100 --rollout 100 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mn_rollout T=100, $n envs:', round(d['value']/1e6,1), 'M env steps/s, frac of HBM roofline', round(d['roofline']['frac'],4))"
  python bench.py --no-learner --envs $n --steps 300 --warmup 50 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mn_step + mn_reset_done per launch pair, $n envs:', round(d['value']/1e6,1), 'M env steps/s,', round(d['ms_per_step']*1e3,1), 'us per vector step')"
done > $O/rollout_scaling.txt 2>&1
python bench.py --no-learner --envs 4096 --steps 1000 --warmup 250 --rollout 250 --cpu-steps 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('mn_rollout T=250, 4096 envs:', round(d['value']/1e6,1), 'M env steps/s')" >> $O/rollout_scaling.txt
