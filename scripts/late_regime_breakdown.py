import os, sys, time
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.iqn.fused_act import act_context
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
n, dev = 65536, "cuda:0"
for L in (1000, 40):
  for eps in (0.9, 0.05):
    for train in (False, True):
        env = VecMarineNavEnv(n, seed=0, device=dev, precision="f64")
        env.params.max_episode_steps = L
        env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1, learning_starts=0, UPDATE_EVERY=4 if train else 10 ** 9)
        agent.reset_under_act = True
        obs = env.reset()
        if L < 1000:
            env.set_state(episode_timesteps=np.random.RandomState(0).randint(0, L, size=n))
        for _ in range(60):
            obs = agent.vec_step(env, obs, eps)[0]
        torch.cuda.synchronize()
        env.profile_begin(50); act_context(agent.qnetwork_local).profile_begin(50)
        t0 = time.perf_counter()
        for _ in range(200):
            obs = agent.vec_step(env, obs, eps)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.join_reset()
        r_ms, _ = env.profile_reset_end(); s_ms, _ = env.profile_end(); a_ms, _ = act_context(agent.qnetwork_local).profile_end()
        cnt = []
        for _ in range(8):
            a = agent.act_batch(obs, eps); env.step(a); cnt.append(env.last_done_count()); obs = env.reset_done()
        print(f"L {L:4d} eps {eps:4.2f} train {int(train)}: {1e3 * dt / 200:.4f} ms per step | act {1e3 * a_ms:.1f} us step {1e3 * s_ms:.1f} us reset {1e3 * r_ms:.1f} us | ends per step {sum(cnt) / 8:.0f}", flush=True)
        env.close()
