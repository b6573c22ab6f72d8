# Ablation of the env-tiled shared-tau act kernel: builds libsp_ablT<bits>.so with -DTILED_ABL=<bits> (csrc/iqn_act_tiled.h) and times the act
# call at 65 536 envs.  Results are wrong by construction; only the time is of interest.
#   bash scripts/act_tiled_ablation.sh build   (here)        bash scripts/act_tiled_ablation.sh run   (GPU box)
set -e
D=distributional_rl_navigation_amd
BITS="0 1 2 3 4 8 12 15"
if [ "$1" = build ]; then
  cd $D/csrc
  for a in $BITS; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -ffp-contract=fast -fno-slp-vectorize -DTILED_ABL=$a -c iqn_act.hip -o /tmp/iqn_act_ablT$a.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsp_ablT$a.so mn_reset.o mn_step.o mn_rollout.o mn_capi.o /tmp/iqn_act_ablT$a.o replay.o iqn_train.o dqn_act.o
  done
else
  cp $D/libmarinenav_hip.so /tmp/lib_keep.so
  for a in $BITS; do
    cp $D/libsp_ablT$a.so $D/libmarinenav_hip.so
    echo -n "TILED_ABL=$a  "; python scripts/act_shared_ab.py 65536 1 2>/dev/null | tail -1
  done
  cp /tmp/lib_keep.so $D/libmarinenav_hip.so
fi
