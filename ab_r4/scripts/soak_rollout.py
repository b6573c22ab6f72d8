"""Soak of mn_rollout: many launches of T in-kernel steps with in-kernel resets; invariants checked on every trace."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
from distributional_rl_navigation_amd.train_iqn import TRAINING_SCHEDULE

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
launches = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
T = 100
env = VecMarineNavEnv(n, seed=0, schedule=TRAINING_SCHEDULE, timestep_scale=3_000_000 / (launches * T), device="cuda:0")
env.reset()
t0 = time.time(); episodes = 0; succ = 0
for k in range(launches):
    out = env.rollout(T, action_seed=1, first_step=k * T, trace=("obs", "reward", "done", "info"))
    if k % 50 == 0 or k == launches - 1:
        assert torch.isfinite(out["obs"]).all() and torch.isfinite(out["reward"]).all()
        assert ((out["info"] != 0) == out["done"].bool()).all()
        episodes += int(out["done"].sum()); succ += int((out["info"] == 4).sum())
torch.cuda.synchronize()
dt = time.time() - t0
s, ep, tot = env.get_state()
assert (tot == launches * T).all() and (ep <= 1001).all() and (ep >= 0).all()
w = env.get_worlds(0, 256)
print(f"{n} envs x {launches} launches x {T} steps = {n * launches * T:.3g} env steps in {dt:.1f} s = {n * launches * T / dt / 1e6:.0f} M env steps/s; "
      f"sampled traces: {episodes} episodes ended ({succ} at the goal); counters exact; final curriculum stage worlds: "
      f"cores {max(x['n_cores'] for x in w)}, obstacles {max(x['n_obs'] for x in w)}; torch memory {torch.cuda.memory_allocated() / 1e6:.0f} MB")
