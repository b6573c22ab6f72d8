"""The reference's full comparison (run_experiments.py:213-282: 8 policies x 500 worlds, exp_setup_5) on one GPU with the
reference's shipped IQN and DQN checkpoints (tests/golden/pretrained_*).  python scripts/experiment_sweep.py [num] [n_obs] [n_cores]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
from distributional_rl_navigation_amd.dqn import DQNPolicy
from distributional_rl_navigation_amd.experiments import ALL_POLICIES, run_experiment
from distributional_rl_navigation_amd.iqn.agent import IQNAgent

num = int(sys.argv[1]) if len(sys.argv) > 1 else 500
n_obs = int(sys.argv[2]) if len(sys.argv) > 2 else 10
n_cores = int(sys.argv[3]) if len(sys.argv) > 3 else 8
G = os.path.join(ROOT, "tests", "golden")
agent = IQNAgent(26, 9, device="cuda:0", seed=0, BUFFER_SIZE=1024)
agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
dqn = DQNPolicy.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"), device="cuda:0")
run_experiment(agent, n_obs, n_cores, num=8, policies=ALL_POLICIES, dqn=dqn)      # warm-up
torch.cuda.synchronize(); t0 = time.perf_counter()
res, _ = run_experiment(agent, n_obs, n_cores, num=num, seed=15, policies=ALL_POLICIES, dqn=dqn)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f"# run_experiment(pretrained IQN seed_3 + DQN seed_3, n_obs={n_obs}, n_cores={n_cores}, num={num}, seed=15): "
      f"{num} worlds x {len(ALL_POLICIES)} policies = {num * len(ALL_POLICIES)} episodes side by side on one MI355X: {dt:.2f} s wall-clock")
for name, r in res.items():
    ok = np.array(r["success"])
    print(f"{name:13s} success {ok.mean():.2f}  out_of_area {np.mean(r['out_of_area']):.2f}  "
          f"avg_time {np.mean(np.array(r['time'])[ok]) if ok.any() else float('nan'):.1f}  "
          f"avg_energy {np.mean(np.array(r['energy'])[ok]) if ok.any() else float('nan'):.1f}")
