# Round-4 measurements (run on the GPU box from the repo root): the bench line, rocprofv3 kernel stats of the headline loop with per-env and with
# launch-shared taus, of the training cadence, PMC passes (HBM traffic of the env kernels now that the float64 copies are opt-in, instruction mix of the
# act kernels), the learner alone, the experiment sweep with the classical baselines / DQN on their kernels.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${MEASURE_DIR:-m4}; mkdir -p $O; rm -f $O/pmc_summary.txt $O/pmc_shared_summary.txt
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --shared-taus --cpu-steps 0 --no-also --no-learner-only > $O/bench_shared_taus.json 2> $O/bench_shared_taus.err
python scripts/learner_bench.py 3000 > $O/learner_bench.txt 2>&1
python scripts/experiment_sweep.py > $O/experiment_sweep.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_loop -- python $R/bench.py --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe > $O/prof_loop.log 2>&1
python $R/scripts/prof_summary.py $(find $O/prof_loop -name "*kernel_stats.csv" | head -1) 14 > $O/prof_loop_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_shared -- python $R/bench.py --shared-taus --steps 100 --warmup 20 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe > $O/prof_shared.log 2>&1
python $R/scripts/prof_summary.py $(find $O/prof_shared -name "*kernel_stats.csv" | head -1) 14 > $O/prof_shared_summary.txt
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_g16 -- python $R/bench.py --steps 60 --warmup 10 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe --update-every 1 --grad-steps 16 --eps 0.05 > $O/prof_g16.log 2>&1
python $R/scripts/prof_summary.py $(find $O/prof_g16 -name "*kernel_stats.csv" | head -1) 14 > $O/prof_g16_summary.txt
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 24 --warmup 8 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe --update-every 1 --grad-steps 4 > $O/pmc_$c.log 2>&1
  python $R/scripts/pmc_agg.py $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) >> $O/pmc_summary.txt
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcs_$c -- python $R/bench.py --shared-taus --steps 24 --warmup 8 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe > $O/pmcs_$c.log 2>&1
  python $R/scripts/pmc_agg.py $(find $O/pmcs_$c -name "*counter_collection.csv" | head -1) act >> $O/pmc_shared_summary.txt
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; du -sh $O
