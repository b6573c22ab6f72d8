#!/bin/bash
# The multi-GPU lines of BASELINE configs[3] / configs[4] on ONE node: 1, 2, 4, 8 GPUs (as many as the node has), one process per GPU
# over RCCL, weak scaling (65 536 envs per GPU).  One JSON line per run in $OUT (default gpurun_out/scale):
#   c3_N.json  configs[3]: N x 65 536 envs, independent IQN learner per GPU, no collective on the data path
#   c4_N.json  configs[4]: the same envs, ONE shared IQN (RCCL all-reduce of the 143 KB gradient bucket per gradient step; line carries
#              `all_reduce_ms`), CVaR(0.5) action selection; bench cadence (1 gradient step per 4 vector steps)
#   c4t_N.json configs[4] at the cadence that trains (16 gradient steps per vector step), gradient steps of an event as one hipGraph
#   c4m_N.json / c4mt_N.json  the same two with the MAILBOX exchange (--exchange mailbox: every rank's gradient step publishes into and gathers from the ranks'
#              IPC-mapped mailboxes inside its own launch -- no collective; iqn/mailbox.py)
# usage: bash scripts/scale.sh [max_gpus]      (RANKS_PER_GPU=2 bash scripts/scale.sh 2: two ranks on ONE GPU over gloo -- the N > 1 code path without a node)
set -u
cd "$(dirname "$0")/.."
OUT=${OUT:-gpurun_out/scale}; mkdir -p "$OUT"
export HSA_ENABLE_IPC_MODE_LEGACY=0
RPG=${RANKS_PER_GPU:-1}
HAVE=$(python -c "import torch; print(torch.cuda.device_count() * $RPG)")
MAX=${1:-$HAVE}
PORT=29511
for N in 1 2 4 8; do
  [ "$N" -gt "$MAX" ] && break
  [ "$N" -gt "$HAVE" ] && break
  run() { # name, extra args
    local name=$1; shift
    if [ "$N" -eq 1 ]; then python bench.py --gpus 1 --steps 200 --warmup 20 --cpu-steps 0 --no-also --no-learner-only "$@" > "$OUT/${name}_$N.json" 2> "$OUT/${name}_$N.err"
    else python -m torch.distributed.run --nnodes=1 --nproc-per-node "$N" --master-addr 127.0.0.1 --master-port $PORT bench.py --gpus "$N" --ranks-per-gpu "$RPG" --steps 200 --warmup 20 --cpu-steps 0 --no-also --no-learner-only "$@" > "$OUT/${name}_$N.json" 2> "$OUT/${name}_$N.err"; fi
    PORT=$((PORT + 1))
    python - "$OUT/${name}_$N.json" "$name" "$N" <<'PY'
import json, sys
try:      # (a leg that failed must not take the summary of the others with it)
    d = json.loads(open(sys.argv[1]).read().strip().splitlines()[-1])
    print(f"{sys.argv[2]:4s} N={d['n_gpus']}: {d['value'] / 1e6:8.1f} M env steps/s  {d['ms_per_step']:.4f} ms/step  grad-steps/s {d['grad_steps_per_sec']:9.1f}  "
          f"all_reduce_ms {d.get('all_reduce_ms')}  timeouts {d.get('timeouts')}  act launch_ms {d['roofline'].get('launch_ms')}  act frac_algorithmic {d['roofline'].get('frac_algorithmic')}  env-step HBM frac {d['roofline_env_step']['frac']:.3f}")
except Exception as e:
    err = sys.argv[1][:-5] + ".err"
    try:
        last = [l for l in open(err).read().strip().splitlines() if l.strip()][-1][:200]
    except Exception:
        last = "no stderr"
    print(f"{sys.argv[2]:4s} N={sys.argv[3]}: FAILED ({type(e).__name__}); last line of {err}: {last}")
PY
  }
  run c3
  run c4 --shared-learner --cvar 0.5
  run c4t --shared-learner --cvar 0.5 --update-every 1 --grad-steps 16 --eps 0.05 --graph-train
  run c4m --shared-learner --cvar 0.5 --exchange mailbox
  run c4mt --shared-learner --cvar 0.5 --exchange mailbox --update-every 1 --grad-steps 16 --eps 0.05
done
