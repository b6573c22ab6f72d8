#!/bin/bash
# 4-seed A/B of the training run of scripts/train_headline.py (65 536 envs, 16 gradient steps per vector step, the reference's learner
# budget of 93 760 batch-256 steps): acting with per-env taus (default) vs launch-shared taus (IQNAgent.shared_taus).  Alternated on ONE GPU.
cd "$(dirname "$0")/.."
for seed in 100 101 102 103; do
  for mode in "" "--shared-taus"; do
    python scripts/train_headline.py --update-every 1 --grad-steps 16 --seconds 40 --evals 4 --seed $seed $mode 2>&1 | grep -v "amdgpu.ids"
  done
done
