cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for c in SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_MISC SQ_ACTIVE_INST_SCA SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_LDS GRBM_GUI_ACTIVE; do
  rm -rf /tmp/pmc_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c --output-format csv -- python $R/scripts/act_only.py 2 16 > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  [ -n "$f" ] && python $R/scripts/pmc_agg.py $f act
done
