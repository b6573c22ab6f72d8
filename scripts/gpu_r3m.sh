cd $GRAFT_REPO_ROOT
O=gpurun_out/r3m; mkdir -p $O
for v in "--halves 2 --no-also --no-learner-only" "--halves 2 --no-also" "--halves 1 --no-also --no-learner-only"; do
timeout 900 python bench.py --cpu-steps 0 $v > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('$v', d['value'], d['ms_per_step'], d['config']['eps'])"
done
timeout 900 python bench.py --cpu-steps 0 --no-learner-only > $O/b.json 2> $O/b.err; python -c "
import json; d=json.load(open('$O/b.json')); print('default no-learner-only', d['value'], d['also']['two_halves_two_streams'])"
