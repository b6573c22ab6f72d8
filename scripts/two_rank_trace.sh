#!/bin/bash
# rocprofv3 kernel traces of TWO ranks on ONE GPU for the configs[3] / configs[4] legs (scripts/scale.sh's N = 2 rehearsal), and what each rank's kernels overlap with
# (scripts/two_rank_timeline.py).  usage: bash scripts/two_rank_trace.sh <out dir>
R=${GRAFT_REPO_ROOT:-$(cd "$(dirname "$0")/.." && pwd)}; O=${1:-$R/gpurun_out/two_rank}; mkdir -p $O; O=$(cd $O && pwd)
export HSA_ENABLE_IPC_MODE_LEGACY=0
cd /tmp; export TMPDIR=/tmp
PORT=29611
for leg in "c3" "c4 --shared-learner --cvar 0.5" "c4m --shared-learner --cvar 0.5 --exchange mailbox"; do
  set -- $leg; name=$1; shift
  rocprofv3 --kernel-trace --output-format csv -d $O/$name -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port $PORT \
      $R/bench.py --gpus 2 --ranks-per-gpu 2 --steps 60 --warmup 20 --windows 2 --cpu-steps 0 --no-also --no-learner-only --no-clock-probe "$@" > $O/$name.json 2> $O/$name.err
  PORT=$((PORT + 1))
  python $R/scripts/two_rank_timeline.py $O/$name $name
  find $O/$name -name "*.db" -delete
done
