cd $GRAFT_REPO_ROOT
O=gpurun_out/r3l; mkdir -p $O
timeout 2400 python -m pytest tests -x -q -m gpu 2>&1 | tail -15 > $O/pytest_gpu.log; cat $O/pytest_gpu.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -2
timeout 900 python bench.py > $O/bench.json 2> $O/bench.err; tail -3 $O/bench.err; python -c "
import json; d=json.load(open('$O/bench.json')); print(d['value'], d['ms_per_step'], d['learner_only_grad_steps_per_sec_per_gpu'], d['roofline']['frac_algorithmic']); print({k:(v.get('value'), v.get('grad_steps_per_sec')) for k,v in d['also'].items() if 'value' in v}); print(d['also']['shared_learner_ws1'])"
