"""Per-env-tau split-f16 act kernel vs the launch-shared-tau kernel, alternated on ONE GPU (the boxes of the pool differ by 10 %):
HIP-event time per 65 536-env launch through the library's own profiling hooks (kernel + its preparation launch)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.model import ObsEncoder
from distributional_rl_navigation_amd.iqn.fused_act import ActRng, act_context, fused_act
n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 4
net = ObsEncoder(26, 9, seed=1, device="cuda:0")
ctx = act_context(net)
obs = torch.randn(n, 26, device="cuda:0") * 5.0
rng = ActRng(1, "cuda:0")
def run(shared, calls=40):
    for _ in range(5): fused_act(net, obs, 0.05, 1.0, rng=rng, shared_taus=shared)
    torch.cuda.synchronize()
    ctx.profile_begin(calls)
    for _ in range(calls): fused_act(net, obs, 0.05, 1.0, rng=rng, shared_taus=shared)
    ms, k = ctx.profile_end()
    return ms * 1e3
for r in range(rounds):
    a = run(False); b = run("wave"); c = run("tiled")
    print(f"round {r}: per-env taus {a:7.1f} us   shared taus, wavefront per env {b:7.1f} us ({b / a:.3f})   shared taus, env-tiled {c:7.1f} us ({c / a:.3f})"
          f"   ({n} envs, prep launches included)", flush=True)
