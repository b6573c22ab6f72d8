# One round's measurement session (run on the GPU box from the repo root; outputs under gpurun_out/$MEASURE_DIR, summaries copied into profiles/ by
# scripts/assemble_profiles.py): the bench line; rocprofv3 --kernel-trace --stats of the headline loop (per-env and launch-shared taus) and of the
# training cadence; PMC passes, one counter per run (HBM traffic of the env / act kernels, MFMA-busy cycles, instruction mix); the learner alone by
# launch form; the reset kernel against the number of resets; configs[3] / configs[4]
# lines at N = 1 and with two ranks on the one GPU; the experiment sweep.
R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/${MEASURE_DIR:-m6}; mkdir -p $O; rm -f $O/pmc_summary.txt $O/pmc_shared_summary.txt
cd $R
python bench.py > $O/bench_default.json 2> $O/bench_default.err
python bench.py --shared-taus --cpu-steps 0 --no-also --no-learner-only > $O/bench_shared_taus.json 2> $O/bench_shared_taus.err
python scripts/learner_bench.py 3000 > $O/learner_bench.txt 2>&1
python scripts/reset_scaling.py f64 > $O/reset_scaling.txt 2>&1
python scripts/reset_under_act_ab.py 200 > $O/reset_under_act_ab.txt 2>&1
LS=1000,64,32,24,16,12,10,9,8,7 python scripts/reset_under_act_threshold.py 100 > $O/reset_under_act_threshold.txt 2>&1
PROFILE=1 LS=1000,28 python scripts/reset_under_act_threshold.py 100 > $O/reset_under_act_profile.txt 2>&1
python scripts/soak.py 20000 1 4 > $O/soak.txt 2>&1
python scripts/experiment_sweep.py > $O/experiment_sweep.txt 2>&1
OUT=$O/scale bash scripts/scale.sh 1 > $O/scale_n1.txt 2>&1
OUT=$O/scale2 RANKS_PER_GPU=2 bash scripts/scale.sh 2 > $O/scale_two_ranks_one_gpu.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 -L 2>/dev/null | grep -i "TCC_EA" | head -60 > $O/counters_tcc_ea.txt
prof() { # name, bench args
  local name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python $R/bench.py --cpu-steps 0 --no-learner-only --no-also --no-clock-probe "$@" > $O/prof_$name.log 2>&1
  python $R/scripts/prof_summary.py $(find $O/prof_$name -name "*kernel_stats.csv" | head -1) 14 > $O/prof_${name}_summary.txt
}
prof loop --steps 100 --warmup 20
prof loop_reset_in_front --steps 100 --warmup 20 --reset-in-front
prof shared --shared-taus --steps 100 --warmup 20
prof g16 --steps 60 --warmup 10 --update-every 1 --grad-steps 16 --eps 0.05
for form in 1 2 3; do
  rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_learner_$form -- python $R/scripts/learner_prof.py $form > $O/prof_learner_$form.log 2>&1
  python $R/scripts/prof_summary.py $(find $O/prof_learner_$form -name "*kernel_stats.csv" | head -1) 6 > $O/prof_learner_${form}_summary.txt
done
for c in FETCH_SIZE WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -- python $R/bench.py --steps 24 --warmup 8 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe --update-every 1 --grad-steps 4 > $O/pmc_$c.log 2>&1
  python $R/scripts/pmc_agg.py $(find $O/pmc_$c -name "*counter_collection.csv" | head -1) >> $O/pmc_summary.txt
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmcs_$c -- python $R/bench.py --shared-taus --steps 24 --warmup 8 --cpu-steps 0 --no-learner-only --no-also --no-clock-probe > $O/pmcs_$c.log 2>&1
  python $R/scripts/pmc_agg.py $(find $O/pmcs_$c -name "*counter_collection.csv" | head -1) act >> $O/pmc_shared_summary.txt
done
find $O -name "*.db" -delete; find $O -name "*.csv" -size +1M -delete; du -sh $O
