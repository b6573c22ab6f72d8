cd $GRAFT_REPO_ROOT
O=gpurun_out/r3i; mkdir -p $O
timeout 600 python -m pytest tests/test_split_loop_gpu.py -x -q -m gpu 2>&1 | tail -25 > $O/pytest_split.log; cat $O/pytest_split.log
timeout 900 python bench.py --cpu-steps 0 > $O/bench_h1.json 2> $O/bench_h1.err; tail -3 $O/bench_h1.err; python -c "
import json; d=json.load(open('$O/bench_h1.json')); print(d['value'], d['ms_per_step'], d['learner_only_grad_steps_per_sec_per_gpu']); print(json.dumps(d['also'], indent=1))"
timeout 900 python bench.py --cpu-steps 0 --halves 2 --no-also > $O/bench_h2.json 2> $O/bench_h2.err; tail -3 $O/bench_h2.err; python -c "
import json; d=json.load(open('$O/bench_h2.json')); print(d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline_env_step']['launch_ms'])"
