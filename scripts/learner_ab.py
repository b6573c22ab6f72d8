"""A / B of library builds on ONE GPU: back-to-back gradient steps per launch form.  usage: python scripts/learner_ab.py [reps]"""
import os, sys, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from distributional_rl_navigation_amd import _capi
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
dev = "cuda:0"
ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1)
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ag.memory.add_batch(torch.randn(n, 26, device=dev, generator=g), torch.randint(0, 9, (n,), device=dev, generator=g), torch.randn(n, device=dev, generator=g),
                    torch.randn(n, 26, device=dev, generator=g), (torch.rand(n, device=dev, generator=g) < 0.05).float())
out = []
for launches in (3, 2, 1):
    ag.two_launch_step, ag.one_launch_step = launches != 3, launches == 1
    G = 1
    for _ in range(20):
        ag.train_steps_from_memory(G) if G > 1 else ag.train_from_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps // G):
        ag.train_steps_from_memory(G) if G > 1 else ag.train_from_memory()
    torch.cuda.synchronize()
    out.append("%s %.2f" % ("G16" if G > 1 else f"{launches}L", 1e6 * (time.perf_counter() - t0) / (reps // G * G)))
print(" | ".join(out), "us per step; timeouts", ag._fused.timeouts(), flush=True)
