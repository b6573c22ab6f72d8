cd $GRAFT_REPO_ROOT
O=gpurun_out/r3n; mkdir -p $O
timeout 600 python -m pytest tests/test_multigpu_paths_gpu.py -x -q -m gpu -k "graphed" 2>&1 | tail -25 > $O/pytest_graph.log; cat $O/pytest_graph.log
OUT=$O/scale timeout 900 bash scripts/scale.sh 1 2>&1 | tail -5
timeout 300 python bench.py --cpu-steps 0 --no-also --no-learner-only --update-every 1 --grad-steps 16 --eps 0.05 > $O/t16.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/t16.json')); print('eager 16/step', d['value'], d['grad_steps_per_sec'])"
timeout 300 python bench.py --cpu-steps 0 --no-also --no-learner-only --update-every 1 --grad-steps 16 --eps 0.05 --graph-train > $O/t16g.json 2>/dev/null; python -c "
import json; d=json.load(open('$O/t16g.json')); print('graph 16/step', d['value'], d['grad_steps_per_sec'])"
