#!/bin/bash
# third pass: the small-env corner the first two passes point to (fewer envs per GPU = more vector steps for the same learner budget and env-step count)
OUT=${1:-gpurun_out/lc3}; mkdir -p $OUT
run() { python scripts/learning_curve.py "${@:3}" --evals 15 --envs $1 --replay $2 --tag "envs $1 ring $2 ${*:3}" > $OUT/lc_$1_$2_$3$4.txt 2> $OUT/lc_$1_$2_$3$4.err; tail -6 $OUT/lc_$1_$2_$3$4.txt | head -2; }
run 4096 100000 --first-seed 12
run 16384 100000 --first-seed 12
run 8192 100000 --seeds 12
run 4096 300000 --seeds 12
run 2048 100000 --seeds 12
