"""How close are the fused act kernel and eager float32 PyTorch to a float64 evaluation of the same net?"""
import os, sys, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.model import ObsEncoder
from distributional_rl_navigation_amd.iqn.fused_act import fused_qvals
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
net = ObsEncoder.load(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "pretrained_IQN_seed3"), "cuda:0")
env = VecMarineNavEnv(16384, seed=1, device="cuda:0"); obs = env.reset().clone()
taus = torch.rand(16384, 32, device="cuda:0")
with torch.no_grad():
    q32 = net.get_qvals(obs, 1.0, taus=taus)
    net64 = copy.deepcopy(net).double()
    net64.pis = net64.pis.double()
    q64 = net64.get_qvals(obs.double(), 1.0, taus=taus.double())
qf = fused_qvals(net, obs, 1.0, taus=taus)
for name, q in (("eager f32 PyTorch", q32), ("fused HIP kernel ", qf)):
    e = (q.double() - q64).abs()
    print(f"{name}: max |err| vs f64 = {e.max().item():.3e}   mean = {e.mean().item():.3e}   (|Q| up to {q64.abs().max().item():.1f})")
print(f"fused vs eager f32: max |diff| = {(qf - q32).abs().max().item():.3e};  greedy action agreement = {(qf.argmax(1) == q32.argmax(1)).float().mean().item():.5f}")
