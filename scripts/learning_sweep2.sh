#!/bin/bash
# second pass of scripts/learning_sweep.sh around its best cell (16 384 envs, 1 M-row ring): neighbours, and 12 more seeds of the cell itself
OUT=${1:-gpurun_out/lc2}; mkdir -p $OUT
run() { python scripts/learning_curve.py "${@:3}" --evals 15 --envs $1 --replay $2 --tag "envs $1 ring $2 ${*:3}" > $OUT/lc_$1_$2_$3$4.txt 2> $OUT/lc_$1_$2_$3$4.err; tail -6 $OUT/lc_$1_$2_$3$4.txt | head -5; }
run 16384 1000000 --first-seed 12
run 16384 300000 --seeds 12
run 16384 2000000 --seeds 12
run 8192 1000000 --seeds 12
run 32768 1000000 --seeds 12
run 32768 300000 --seeds 12
