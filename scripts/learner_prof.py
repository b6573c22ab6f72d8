"""For rocprofv3: 600 back-to-back gradient steps in one launch form (argv[1] = 3, 2, 1 launches per step)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
form = int(sys.argv[1]) if len(sys.argv) > 1 else 1
dev = "cuda:0"
ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1)
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ag.memory.add_batch(torch.randn(n, 26, device=dev, generator=g), torch.randint(0, 9, (n,), device=dev, generator=g), torch.randn(n, device=dev, generator=g),
                    torch.randn(n, 26, device=dev, generator=g), (torch.rand(n, device=dev, generator=g) < 0.05).float())
ag.two_launch_step, ag.one_launch_step = form != 3, form == 1
for _ in range(600):
    ag.train_from_memory()
torch.cuda.synchronize()
