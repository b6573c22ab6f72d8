#!/bin/bash
# The structural knobs of the batched training run against the reference's learning curve: replay ring size x envs per GPU, same learner
# budget (93 760 gradient steps of batch 256), 12 seeds each.  usage: scripts/learning_sweep.sh <out dir> [seeds]
OUT=${1:-gpurun_out/lc}; SEEDS=${2:-12}
mkdir -p $OUT
for cfg in "65536 100000" "65536 1000000" "65536 4000000" "65536 16000000" "16384 100000" "16384 1000000" "16384 4000000" "4096 100000" "4096 1000000" "4096 4000000"; do
  set -- $cfg
  python scripts/learning_curve.py --seeds $SEEDS --evals 15 --envs $1 --replay $2 --tag "envs $1 ring $2" > $OUT/lc_$1_$2.txt 2> $OUT/lc_$1_$2.err
  tail -6 $OUT/lc_$1_$2.txt | head -5
done
