#!/bin/bash
# A/B on one GPU, alternating: mn_reset_done next to the gradient steps (second stream, MN_OVERLAP_RESET=1) vs in front of them (default).
# Prints ms per vector step of the main loop (1 gradient step / 4 vector steps) and of the training cadence (16 per step).
mkdir -p gpurun_out/r4b
for rep in 1 2 3; do
  for ov in 1 0; do
    MN_OVERLAP_RESET=$ov python bench.py --no-clock-probe --cpu-steps 0 --steps 200 2>/dev/null | python -c "
import json, sys
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
a = d['also']
print('overlap=$ov  value %.1f M (%.4f ms)   train_cadence %.1f M, %.0f grad-steps/s (%.4f ms)   shared-tau cadence %.4f ms' % (
    d['value'] / 1e6, d['ms_per_step'], a['train_cadence']['value'] / 1e6, a['train_cadence']['grad_steps_per_sec'], a['train_cadence']['ms_per_step'],
    a['train_cadence_shared_taus']['ms_per_step']))"
  done
done
