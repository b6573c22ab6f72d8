"""Micro-benchmark of the fused act kernel at several batch sizes (prologue vs per-env cost)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.model import ObsEncoder
from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
net = ObsEncoder(26, 9, seed=1, device="cuda:0")
act_context(net).set_variant(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
for n in (2048, 4096, 8192, 16384, 65536, 131072):
    obs = torch.randn(n, 26, device="cuda:0"); taus = torch.rand(n, 32, device="cuda:0")
    for _ in range(5): fused_act(net, obs, 0.0, 1.0, taus=taus)
    torch.cuda.synchronize()
    s = torch.cuda.Event(enable_timing=True); e = torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): fused_act(net, obs, 0.0, 1.0, taus=taus)
    e.record(); torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 20
    print(f"n={n:7d}  {ms*1e3:8.1f} us/call  {n/ms/1e3:8.2f} M env/s  {2.003e6*n/ms/1e9:7.1f} TFLOP/s")
