"""Does the GPU interleave the learner's gradient steps (stream L) with an act launch (stream A)?  Times, for several act grids:
act alone, 16 gradient steps alone, both issued together on two streams.  usage: python scripts/overlap_probe.py"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act, ActRng
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv

dev = "cuda:0"
n = 65536
ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1)
g = torch.Generator(device=dev); g.manual_seed(0)
m = 100_000
ag.memory.add_batch(torch.randn(m, 26, device=dev, generator=g), torch.randint(0, 9, (m,), device=dev, generator=g),
                    torch.randn(m, device=dev, generator=g), torch.randn(m, 26, device=dev, generator=g),
                    (torch.rand(m, device=dev, generator=g) < 0.05).float())
actor = IQNAgent(26, 9, device=dev, seed=2)           # its own network: the learner does not invalidate its weight image
obs = torch.randn(n, 26, device=dev, generator=g) * 5
rng = ActRng(5, dev)
env = VecMarineNavEnv(n // 2, seed=0, device=dev, precision="f64")
env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
env.reset()
acts = torch.randint(0, 9, (n // 2,), device=dev, dtype=torch.int32, generator=g)
A, L, E = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
G = 16


def act():
    with torch.cuda.stream(A):
        fused_act(actor.qnetwork_local, obs, 0.05, 1.0, rng=rng)


def learn():
    with torch.cuda.stream(L):
        for _ in range(G):
            ag.train_from_memory()


def envstep():
    with torch.cuda.stream(E):
        env.step(acts)
        env.reset_done()


def timed(fns, reps=30):
    for _ in range(3):
        for f in fns:
            f()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        for f in fns:
            f()
    torch.cuda.synchronize()
    return 1e6 * (time.perf_counter() - t0) / reps


learn(); torch.cuda.synchronize()
for grid in (0, 512, 1024, 2048, 4096):
    act_context(actor.qnetwork_local).set_grid(grid)
    ta, tl, te = timed([act]), timed([learn]), timed([envstep])
    tal, tae, tale = timed([act, learn]), timed([act, envstep]), timed([act, learn, envstep])
    print(f"act grid {grid:5d}: act {ta:7.1f}  16 grad steps {tl:7.1f}  env(32768) step+reset {te:6.1f} | act||learn {tal:7.1f} (sum {ta + tl:7.1f})  "
          f"act||env {tae:7.1f} (sum {ta + te:7.1f})  all three {tale:7.1f} (sum {ta + tl + te:7.1f})  us", flush=True)
