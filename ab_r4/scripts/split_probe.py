"""SplitBatchLoop variants: which of its orderings cost time?  usage: python scripts/split_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.iqn.overlap import SplitBatchLoop
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
dev = "cuda:0"; n = 65536
def mk(k, first):
    e = VecMarineNavEnv(k, seed=0, first_index=first, device=dev, precision="f64")
    e.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    return e
def run(label, H, grid, order, update_every, reps=200, gsteps=1, lstream=False):
    ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1, learning_starts=0, UPDATE_EVERY=update_every)
    ag.grad_steps_per_update = gsteps
    envs = [mk(n // H, h * (n // H)) for h in range(H)]
    loop = SplitBatchLoop(ag, envs, act_grid=grid, order_appends=order, learner_stream=lstream)
    loop.reset()
    for _ in range(20):
        loop.step(0.9)
    loop.join(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        loop.step(0.9)
    loop.join(); torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    print(f"{label:60s}: {1e6 * dt:7.1f} us per vector step ({n / dt / 1e6:6.1f} M env steps/s)", flush=True)
    loop.close(); torch.cuda.synchronize()
    for e in envs:
        e.close()
for rnd in range(2):
    run("H=1 (side stream), train every 4", 1, 0, False, 4)
    run("H=2 grid 1024, no append order, never train", 2, 1024, False, 10**9)
    run("H=2 grid 1024, append order events, never train", 2, 1024, True, 10**9)
    run("H=2 grid 1024, no append order, train every 4", 2, 1024, False, 4)
    run("H=2 grid 1024, append order events, train every 4", 2, 1024, True, 4)
    run("H=2 grid 1024, append order, 16 grad steps per step", 2, 1024, True, 1, 100, 16)
    run("H=1, 16 grad steps per step", 1, 0, True, 1, 100, 16)
    run("H=2 grid 1024, learner stream, 16 grad steps per step", 2, 1024, True, 1, 100, 16, True)
    run("H=2 grid 2048, learner stream, 16 grad steps per step", 2, 2048, True, 1, 100, 16, True)
    run("H=1 grid 1024, learner stream, 16 grad steps per step", 1, 1024, True, 1, 100, 16, True)
    run("H=2 grid 1024, learner stream, train every 4", 2, 1024, True, 4, 200, 1, True)
