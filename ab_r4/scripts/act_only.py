"""N act calls of one kernel variant at 65 536 envs -- the workload for rocprofv3 kernel-trace / PMC passes on the act kernel."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.model import ObsEncoder
from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
variant = int(sys.argv[1]) if len(sys.argv) > 1 else 0
calls = int(sys.argv[2]) if len(sys.argv) > 2 else 20
n = int(sys.argv[3]) if len(sys.argv) > 3 else 65536
net = ObsEncoder(26, 9, seed=1, device="cuda:0")
act_context(net).set_variant(variant)
obs = torch.randn(n, 26, device="cuda:0") * 5.0; taus = torch.rand(n, 32, device="cuda:0")
for _ in range(calls): fused_act(net, obs, 0.0, 1.0, taus=taus)
torch.cuda.synchronize()
