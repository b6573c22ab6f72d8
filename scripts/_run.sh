cd $GRAFT_REPO_ROOT
D=distributional_rl_navigation_amd
timeout 900 python -m pytest tests/test_act_split_gpu.py tests/test_act_ctx_gpu.py tests/test_iqn_gpu.py tests/test_agent_gpu.py -x -q -m gpu 2>&1 | tail -2
cp $D/libmarinenav_hip.so /tmp/new.so
for rep in 1 2 3; do
for v in new base; do
  if [ $v = new ]; then cp /tmp/new.so $D/libmarinenav_hip.so; else cp $D/libmn_base.so $D/libmarinenav_hip.so; fi
python bench.py --steps 200 --warmup 50 --cpu-steps 0 --no-learner-only --no-also 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$v:', round(d['value']/1e6,2), 'M,', round(d['ms_per_step']*1e3,1), 'us; act launch', round(r['launch_ms']*1e3,1))"
done; done
cp /tmp/new.so $D/libmarinenav_hip.so
