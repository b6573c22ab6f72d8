cd $GRAFT_REPO_ROOT
for L in 0 2 4 8; do
timeout 300 python bench.py --cpu-steps 0 --no-also --no-learner-only --lanes $L 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f64 loop lanes $L', round(d['value']/1e6,1), d['ms_per_step'], 'step kernel ms', d['roofline_env_step']['launch_ms'])"
done
for L in 0 4 8; do
timeout 300 python bench.py --cpu-steps 0 --no-learner --precision f64 --lanes $L --steps 400 --warmup 50 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('f64 step-only lanes $L', round(d['value']/1e6,1), 'step kernel ms', d['roofline']['launch_ms'])"
done
