"""Episode resets in front of the act kernel vs under it (IQNAgent.reset_under_act) against the number of episodes that end per vector step:
65 536 envs, float64, act + step + append + reset, no gradient steps; `max_episode_steps` = L makes 65 536 / L envs finish per step.
usage: python scripts/reset_under_act_ab.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.iqn.fused_act import late_timeouts
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
n, dev = 65536, "cuda:0"
for L in (1000, 128, 64, 32, 16, 8):
    row = []
    for under in (False, "auto", "always", False, "auto", "always"):
        env = VecMarineNavEnv(n, seed=0, device=dev, precision="f64")
        env.params.max_episode_steps = L
        env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1, learning_starts=0, UPDATE_EVERY=10 ** 9)
        agent.reset_under_act = bool(under)
        if under == "always":
            env.set_reset_under_act_max(2 ** 31 - 1)
        obs = env.reset()
        # warm-up (the library's rule needs to have seen a few reset launches)
        for _ in range(60):
            obs = agent.vec_step(env, obs, 1.0)[0]
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            obs = agent.vec_step(env, obs, 0.9)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.join_reset()
        cnt = []
        for _ in range(8):
            a = agent.act_batch(obs, 0.9)
            env.step(a); cnt.append(env.last_done_count()); obs = env.reset_done()
        row.append((1e3 * dt / steps, sum(cnt) / len(cnt), late_timeouts(agent.qnetwork_local)))
        env.close()
    print(f"max_episode_steps {L:5d}: ~{row[0][1]:7.0f} resets per vector step | in front {row[0][0]:.4f} {row[3][0]:.4f} ms | library's rule (decaying peak <= 5000: under) {row[1][0]:.4f} {row[4][0]:.4f} ms"
          f" | always under the act kernel {row[2][0]:.4f} {row[5][0]:.4f} ms | late-row timeouts {row[1][2] + row[2][2] + row[4][2] + row[5][2]}", flush=True)
