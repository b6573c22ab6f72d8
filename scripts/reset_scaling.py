"""mn_reset_kernel (one wavefront per environment) against the number of environments it resets in one launch: HIP-event time of `reset(mask)` for k of 65 536
envs.  Flat up to one wave per SIMD (1 024) and two (2 048): the kernel is the dependent chain of ONE episode start (MT19937 block in, rejection loops, tables out);
beyond that it scales with the number of waves a SIMD runs in sequence.  usage: python scripts/reset_scaling.py [precision]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
prec = sys.argv[1] if len(sys.argv) > 1 else "f64"
n = 65536
env = VecMarineNavEnv(n, seed=0, device="cuda:0", precision=prec)
env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
env.reset()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
print(f"# mn_reset_kernel<{prec}>, 8 cores / 10 obstacles, 2 048 B of algorithmic traffic per reset (bench.RESET_BYTES)")
for k in (1, 64, 256, 512, 1024, 2048, 4096, 8192, 16384, 65536):
    mask = torch.zeros(n, dtype=torch.uint8, device="cuda:0")
    mask[torch.randperm(n, device="cuda:0")[:k]] = 1      # scattered over the batch, as finished episodes are
    for _ in range(3):
        env.reset(mask)
    torch.cuda.synchronize()
    ts = []
    for _ in range(15):
        e0.record(); env.reset(mask); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    t = ts[len(ts) // 2]
    print(f"{k:6d} resets per launch: {t:8.1f} us (median of 15; mask -> queue launch included)   {2048 * k / t / 1e3:8.2f} GB/s   {k / t:8.2f} resets/us")
