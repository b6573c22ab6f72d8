# Round-2 measurements with the split-f16 act kernel as the default (run on the GPU box from the repo root).
set -x
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/m2; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --act-variant 0 --cpu-steps 0 --no-learner-only > $O/bench_exact_act.json 2> $O/bench_exact_act.err
python bench.py --gpus 1 --shared-learner --cvar 0.5 --cpu-steps 0 --no-learner-only > $O/bench_shared.json 2> $O/bench_shared.err
python bench.py --update-every 1 --grad-steps 16 --cpu-steps 0 --no-learner-only > $O/bench_g16.json 2> $O/bench_g16.err
python bench.py --precision f64 --cpu-steps 0 --no-learner-only > $O/bench_f64env.json 2> $O/bench_f64env.err
python scripts/train_headline.py --update-every 1 --grad-steps 16 --seconds 40 > $O/train_g16.txt 2>&1
python scripts/train_headline.py --update-every 1 --grad-steps 32 --seconds 40 > $O/train_g32.txt 2>&1
( time python -m distributional_rl_navigation_amd.train_iqn --help > /dev/null ) 2> $O/cli_help_time.txt
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_loop -- python $R/bench.py --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only > $O/prof_loop.log 2>&1
python $R/scripts/prof_summary.py $(find $O/prof_loop -name "*kernel_stats.csv" | head -1) 16 > $O/prof_loop_summary.txt
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 30 --warmup 10 --cpu-steps 0 --no-learner-only > $O/pmc_$c.log 2>&1
  python $R/scripts/pmc_agg.py $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) >> $O/pmc_summary.txt
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +2M -delete; du -sh $O
