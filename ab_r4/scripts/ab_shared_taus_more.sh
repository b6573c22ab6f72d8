#!/bin/bash
# eight more seeds of scripts/ab_shared_taus.sh (104..111)
cd "$(dirname "$0")/.."
for seed in 104 105 106 107 108 109 110 111; do
  for mode in "" "--shared-taus"; do
    python scripts/train_headline.py --update-every 1 --grad-steps 16 --seconds 40 --evals 4 --seed $seed $mode 2>&1 | grep -v "amdgpu.ids" | grep "^#.*best\|^# seed"
  done
done
