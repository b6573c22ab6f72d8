# Round-3 measurements (run on the GPU box from the repo root): rocprofv3 kernel stats of the headline loop and of the training cadence,
# PMC passes (HBM traffic, instruction mix) for the loop's kernels and the gradient step, the learner alone, train-to-reference time.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/m3; mkdir -p $O
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --halves 2 --cpu-steps 0 --no-also --no-learner-only > $O/bench_halves2.json 2> $O/bench_halves2.err
python scripts/train_headline.py --update-every 1 --grad-steps 16 --seconds 40 > $O/train_g16.txt 2>&1
python scripts/learner_bench.py 3000 > $O/learner_bench.txt 2>&1
python scripts/train_phase_timing.py 256 300 > $O/train_phases.txt 2>&1
bash scripts/measure_rollout.sh
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_loop -- python $R/bench.py --halves 1 --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only --no-also > $O/prof_loop.log 2>&1
python $R/scripts/prof_summary.py $(find $O/prof_loop -name "*kernel_stats.csv" | head -1) 14 > $O/prof_loop_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g16 -- python $R/bench.py --halves 1 --steps 60 --warmup 10 --cpu-steps 0 --no-learner-only --no-also --update-every 1 --grad-steps 16 --eps 0.05 > $O/prof_g16.log 2>&1
python $R/scripts/prof_summary.py $(find $O/prof_g16 -name "*kernel_stats.csv" | head -1) 14 > $O/prof_g16_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_h2 -- python $R/bench.py --halves 2 --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only --no-also > $O/prof_h2.log 2>&1
python $R/scripts/prof_summary.py $(find $O/prof_h2 -name "*kernel_stats.csv" | head -1) 14 > $O/prof_h2_summary.txt
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --halves 1 --steps 24 --warmup 8 --cpu-steps 0 --no-learner-only --no-also --update-every 1 --grad-steps 4 > $O/pmc_$c.log 2>&1
  python $R/scripts/pmc_agg.py $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) >> $O/pmc_summary.txt
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; du -sh $O
