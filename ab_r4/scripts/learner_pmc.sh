#!/bin/bash
# PMC counters of the gradient step's kernels (learner alone, two launches per step), one counter per rocprofv3 pass.
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cat > /tmp/learner_loop.py <<PY
import sys; sys.path.insert(0, "$R")
import torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
dev = "cuda:0"
ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1)
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ag.memory.add_batch(torch.randn(n, 26, device=dev, generator=g), torch.randint(0, 9, (n,), device=dev, generator=g),
                    torch.randn(n, device=dev, generator=g), torch.randn(n, 26, device=dev, generator=g), (torch.rand(n, device=dev, generator=g) < 0.05).float())
for _ in range(60):
    ag.train_from_memory()
torch.cuda.synchronize()
PY
for c in "$@"; do
  rm -rf /tmp/pmc_$c
  timeout 200 rocprofv3 --pmc $c --kernel-trace -d /tmp/pmc_$c --output-format csv -- python /tmp/learner_loop.py > /tmp/pmc_$c.log 2>&1
  f=$(find /tmp/pmc_$c -name "*counter_collection.csv" | head -1)
  if [ -n "$f" ]; then python $R/scripts/pmc_agg.py $f train; python $R/scripts/pmc_agg.py $f reduce; else echo "$c: no output"; tail -3 /tmp/pmc_$c.log; fi
done
