"""Samples the shader clock / power (rocm-smi) while the fused act kernel runs back to back, to separate "MFMA pipe idle"
from "clock below the 2.4 GHz the peak assumes".  python scripts/act_clock.py"""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.model import ObsEncoder
from distributional_rl_navigation_amd.iqn.fused_act import fused_act

net = ObsEncoder(26, 9, seed=1, device="cuda:0")
n = 65536
obs = torch.randn(n, 26, device="cuda:0"); taus = torch.rand(n, 32, device="cuda:0")
samples = []
stop = False
def sampler():
    while not stop:
        try:
            out = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], capture_output=True, text=True, timeout=10).stdout
            samples.append(out.strip())
        except Exception as e:
            samples.append(f"err {e}")
        time.sleep(0.5)
for _ in range(20): fused_act(net, obs, 0.0, 1.0, taus=taus)
torch.cuda.synchronize()
th = threading.Thread(target=sampler); th.start()
t0 = time.time(); k = 0
while time.time() - t0 < 6.0:
    for _ in range(100): fused_act(net, obs, 0.0, 1.0, taus=taus)
    torch.cuda.synchronize(); k += 100
dt = time.time() - t0
stop = True; th.join()
print(f"{k} launches in {dt:.2f} s -> {dt/k*1e6:.1f} us per launch")
for s in samples[1:6]:
    print(s[:600])
