cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python $R/bench.py --cpu-steps 0 --no-learner-only > $R/gpurun_out/bench_now.json 2>/dev/null
rm -rf /tmp/prof_loop
rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/prof_loop -- python $R/bench.py --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only > /tmp/prof_loop.log 2>&1
f=$(find /tmp/prof_loop -name "*kernel_stats.csv" | head -1)
python $R/scripts/prof_summary.py $f 14 > $R/gpurun_out/prof_loop_split.txt
cat $R/gpurun_out/prof_loop_split.txt
python -c "
import json; d=json.load(open('$R/gpurun_out/bench_now.json')); print('bench', d['value'], d['ms_per_step'], d['roofline']['launch_ms'], d['roofline']['frac'])"
