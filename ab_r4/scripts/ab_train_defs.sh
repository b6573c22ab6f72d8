#!/bin/bash
# A/B of compile-time variants of csrc/iqn_train.hip on one GPU: builds the library with each set of -D flags in turn (in the box's copy of the tree),
# runs scripts/learner_bench.py, restores the default build.  usage: bash scripts/ab_train_defs.sh "-DA=1" "-DA=2" ...
cd $GRAFT_REPO_ROOT/distributional_rl_navigation_amd/csrc
for rep in 1 2; do
  for defs in "$@"; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wall -Wno-unused-function $defs -c iqn_train.hip -o iqn_train.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libmarinenav_hip.so mn_reset.o mn_step.o mn_rollout.o mn_capi.o iqn_act.o replay.o iqn_train.o dqn_act.o
    echo "== $defs"
    (cd $GRAFT_REPO_ROOT && timeout 200 python scripts/learner_bench.py 3000 2>&1 | grep "mode 0" | head -2)
  done
done
