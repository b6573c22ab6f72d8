R=/root/repo; O=$R/gpurun_out/pmcact; mkdir -p $O; cd /tmp; export TMPDIR=/tmp
for v in 0 1; do
  python $R/bench.py --act-variant $v --cpu-steps 0 --no-learner-only >> $O/bench.jsonl 2>/dev/null
  for c in SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE SQ_BUSY_CYCLES SQ_WAIT_INST_ANY; do
    rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/v${v}_$c -- python $R/bench.py --act-variant $v --steps 20 --warmup 5 --cpu-steps 0 --no-learner-only > $O/v${v}_$c.log 2>&1
  done
done
find $O -name "*.db" -delete
