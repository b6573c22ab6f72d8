set -x
R=/root/repo; O=$R/gpurun_out/m1; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --gpus 1 --shared-learner --cvar 0.5 > $O/bench_shared.json 2> $O/bench_shared.err
python bench.py --update-every 1 --grad-steps 16 --cpu-steps 0 --no-learner-only > $O/bench_g16.json 2> $O/bench_g16.err
for T in 0 10 50 100 250; do python bench.py --no-learner --envs 4096 --steps 1000 --warmup 250 --rollout $T --cpu-steps 0 >> $O/configs1.jsonl 2>> $O/configs1.err; done
for T in 0 50; do python bench.py --no-learner --envs 65536 --steps 1000 --warmup 100 --rollout $T --cpu-steps 0 >> $O/configs1.jsonl 2>> $O/configs1.err; done
python bench.py --no-learner --envs 4096 --steps 1000 --warmup 200 --rollout 100 --rollout-trace "" --cpu-steps 0 >> $O/configs1.jsonl 2>> $O/configs1.err
python bench.py --no-learner --envs 16384 --steps 1000 --warmup 200 --rollout 100 --cpu-steps 0 >> $O/configs1.jsonl 2>> $O/configs1.err
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_loop -- python $R/bench.py --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only > $O/prof_loop.log 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_roll -- python $R/bench.py --no-learner --envs 4096 --steps 1000 --warmup 200 --rollout 100 --cpu-steps 0 > $O/prof_roll.log 2>&1
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_loop_$c -- python $R/bench.py --steps 30 --warmup 10 --cpu-steps 0 --no-learner-only > $O/pmc_loop_$c.log 2>&1
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_roll_$c -- python $R/bench.py --no-learner --envs 4096 --steps 400 --warmup 100 --rollout 100 --cpu-steps 0 > $O/pmc_roll_$c.log 2>&1
done
find $O -name "*.db" -delete; du -sh $O
