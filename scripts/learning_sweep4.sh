#!/bin/bash
# fourth pass: 12 more seeds of round 5's cell (a like-for-like 24 against the new default's 24), and n-step 3 at the new default (the reference's n-step window does not
# restart at episode ends, replay_buffer.py:26-41, and its scripts never use it: for completeness of VERDICT r5 item 3's list)
OUT=${1:-gpurun_out/lc4}; mkdir -p $OUT
run() { python scripts/learning_curve.py "${@:3}" --evals 15 --envs $1 --replay $2 --tag "envs $1 ring $2 ${*:3}" > $OUT/lc_$1_$2_$3$4$5$6.txt 2> $OUT/lc_$1_$2_$3$4$5$6.err; tail -6 $OUT/lc_$1_$2_$3$4$5$6.txt | head -2; }
run 65536 100000 --first-seed 12
run 4096 100000 --n-step 3 --seeds 12
