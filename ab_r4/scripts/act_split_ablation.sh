# Ablation of the split-f16 act kernel: builds libmarinenav_hip.so with -DSP_ABL=<bits> (iqn_act_split.h) and times the act call.
# Results are wrong by construction; only the time is of interest.  Run from the repo root:  bash scripts/act_split_ablation.sh build   (here)
#                                                                                          bash scripts/act_split_ablation.sh run     (GPU box)
set -e
D=distributional_rl_navigation_amd
if [ "$1" = build ]; then
  cd $D/csrc
  for a in 0 1 2 4 8 16 3 7 15 31; do
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-function -ffp-contract=fast -DSP_ABL=$a -c iqn_act.hip -o /tmp/iqn_act_abl$a.o 2>/dev/null
    /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../libsp_abl$a.so mn_reset.o mn_step.o mn_rollout.o mn_capi.o /tmp/iqn_act_abl$a.o replay.o iqn_train.o
  done
else
  cp $D/libmarinenav_hip.so /tmp/lib_keep.so
  for a in 0 1 2 4 8 16 3 7 15 31; do
    cp $D/libsp_abl$a.so $D/libmarinenav_hip.so
    echo -n "SP_ABL=$a  "; python scripts/act_micro.py 2 2>/dev/null | grep "n=  65536"
  done
  cp /tmp/lib_keep.so $D/libmarinenav_hip.so
fi
