"""Learner alone: back-to-back fused gradient steps (batch drawn in the launch), per launch mode.
usage: python scripts/learner_bench.py [reps] [batch]"""
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from distributional_rl_navigation_amd import _capi
from distributional_rl_navigation_amd.iqn.agent import IQNAgent

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2000
B = int(sys.argv[2]) if len(sys.argv) > 2 else 256
dev = "cuda:0"
ag = IQNAgent(26, 9, BATCH_SIZE=B, BUFFER_SIZE=100_000, device=dev, seed=1)
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ag.memory.add_batch(torch.randn(n, 26, device=dev, generator=g), torch.randint(0, 9, (n,), device=dev, generator=g),
                    torch.randn(n, device=dev, generator=g), torch.randn(n, 26, device=dev, generator=g),
                    (torch.rand(n, device=dev, generator=g) < 0.05).float())
for mode, launches in ((0, 1), (0, 2), (0, 3), (1, 1), (0, 1), (0, 2), (0, 3)):
    _capi.lib().mn_iqn_train_set_mode(mode)
    # launches per gradient step: 3 = forward / backward, reduction, Adam; 2 = forward / backward, (reduction + clip + Adam) (mn_iqn_train_step);
    # 1 = the same with the reduction + Adam blocks as a third workgroup role of the forward / backward launch (MN_TRAIN_ONE_LAUNCH)
    ag.two_launch_step, ag.one_launch_step = launches < 3, launches == 1
    for _ in range(20):
        ag.train_from_memory()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        ag.train_from_memory()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"mode {mode}, {launches} launch(es) per step: {reps / dt:9.0f} grad-steps/s  ({1e6 * dt / reps:6.2f} us per step), loss {float(ag._fused.loss):.5f}", flush=True)
_capi.lib().mn_iqn_train_set_mode(0)
ag.two_launch_step, ag.one_launch_step = True, False
# the same gradient steps as captured hipGraphs of G steps each (IQNAgent.use_fused_graph)
for G in (16, 64):
    ag.use_fused_graph = True
    for _ in range(3):
        ag.train_steps_from_memory(G)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    k = max(1, reps // G)
    for _ in range(k):
        ag.train_steps_from_memory(G)
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    print(f"hipGraph of {G:3d} steps: {k * G / dt:9.0f} grad-steps/s  ({1e6 * dt / (k * G):6.2f} us per step)", flush=True)
ag.use_fused_graph = False
