"""Where a vector step of mn_rollout goes at BASELINE configs[1] (4 096 envs, random policy, T = 100 per launch): s_memtime stamps of
ONE wavefront (workgroup 0) inside MnLane::step and the rollout loop, from the ABLATION build (libmarinenav_hip_ablation.so -- the only
build with the stamps; the shipped library has none).   python scripts/rollout_phase_timing.py [n_envs] [lanes]"""
import ctypes as C
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd import _capi

ABL = os.path.join(os.path.dirname(_capi.LIB_PATH), "libmarinenav_hip_ablation.so")
L = C.CDLL(ABL)
assert L.mn_build_info() == 1
for name, res, args in _capi.SIGNATURES:
    fn = getattr(L, name); fn.restype, fn.argtypes = res, args
_capi._lib = L
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lanes = int(sys.argv[2]) if len(sys.argv) > 2 else 0
T = 100
env = VecMarineNavEnv(n, seed=0, device="cuda:0", precision="mixed", rollout_lanes=lanes)
env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
env.reset()
for k in range(5):
    env.rollout(T, action_seed=0, first_step=k * T)
torch.cuda.synchronize()
buf = (C.c_ulonglong * 16)()
assert L.mn_debug_rollout_phases(buf, 1) == 0
reps = 20
t0 = time.perf_counter()
for k in range(reps):
    env.rollout(T, action_seed=0, first_step=(5 + k) * T)
torch.cuda.synchronize()
wall = (time.perf_counter() - t0) / (reps * T)
assert L.mn_debug_rollout_phases(buf, 0) == 0
v = [int(x) for x in buf]
steps, resets = v[8], v[9]
tot = sum(v[:8])
# s_memtime ticks -> microseconds: calibrated against the wall clock of the same launches (the stamped wave runs the whole launch)
us_per_tick = wall * steps / tot * 1e6 / 1.0
names = ["action decode, goal distance, sincos", "N = 10 sub-steps (current field + pose integration)", "obstacle rotation, work-list, beam directions",
         "sonar scan + float64 range re-derivation", "reward, termination ladder", "observation row stores",
         "trace writes, done ballot", "in-kernel resets (per step, amortised)"]
print(f"{n} envs, lanes per env {lanes or 'auto'}, {reps} launches of T = {T}: {1e6 * wall:.2f} us per vector step on the wall clock "
      f"({n / wall / 1e6:.0f} M env steps/s); wave 0: {steps} steps stamped, {resets} steps with an in-kernel reset")
print(f"(1 s_memtime tick = {us_per_tick * 1e3:.2f} ns by calibration: the stamped phases cover the wave's whole loop)")
for k, name in enumerate(names):
    print(f"  {name:55s} {v[k] / steps:9.1f} ticks / step  {v[k] / steps * us_per_tick:6.3f} us  {100.0 * v[k] / tot:5.1f} %")
if resets:
    print(f"  (one in-kernel reset: {v[7] / resets * us_per_tick:.2f} us; {resets / steps:.3f} resets per wave and step)")
