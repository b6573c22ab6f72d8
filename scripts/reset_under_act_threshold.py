"""Where the episode resets should run, against a STEADY number of episode ends per vector step: the envs' episode ages are spread uniformly over [0, L) (mn_set_state), so
65 536 / L episodes time out every step (plus the ~150 that end by collision / goal) -- no bursts.  65 536 envs, float64, act + step + append + reset, no gradient steps;
ms per vector step with the resets in front of the act kernel / always under it.  Calibrates mn_reset_done_async's default rule (decaying peak <= 1 200: under).
usage: python scripts/reset_under_act_threshold.py [steps]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.iqn.fused_act import late_timeouts
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 150
n, dev = 65536, "cuda:0"
rng = np.random.RandomState(0)
for L in [int(x) for x in os.environ.get("LS", "1000,400,260,200,160,130,100,64").split(",")]:
    row = []
    for under in (False, True, False, True):
        env = VecMarineNavEnv(n, seed=0, device=dev, precision="f64")
        env.params.max_episode_steps = L
        env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        env.set_reset_under_act_max(2 ** 31 - 1)
        agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device=dev, seed=1, learning_starts=0, UPDATE_EVERY=10 ** 9)
        agent.reset_under_act = under
        obs = env.reset()
        if L < 1000:
            env.set_state(episode_timesteps=rng.randint(0, L, size=n))
        for _ in range(40):
            obs = agent.vec_step(env, obs, 0.9)[0]
        torch.cuda.synchronize()
        prof = os.environ.get("PROFILE") == "1"      # HIP events around the first 50 act / step / reset launches (costs ~18 us of stream time per instrumented step)
        if prof:
            from distributional_rl_navigation_amd.iqn.fused_act import act_context
            env.profile_begin(50); act_context(agent.qnetwork_local).profile_begin(50)
        t0 = time.perf_counter()
        for _ in range(steps):
            obs = agent.vec_step(env, obs, 0.9)[0]
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        env.join_reset()
        if prof:
            r_ms, _ = env.profile_reset_end(); s_ms, _ = env.profile_end(); a_ms, _ = act_context(agent.qnetwork_local).profile_end()
            print(f"    L {L} {'under' if under else 'front'}: act launch {1e3 * a_ms:.1f} us, step kernel {1e3 * s_ms:.1f} us, reset launch {1e3 * r_ms:.1f} us", flush=True)
        cnt = []
        for _ in range(8):
            a = agent.act_batch(obs, 0.9)
            env.step(a); cnt.append(env.last_done_count()); obs = env.reset_done()
        row.append((1e3 * dt / steps, sum(cnt) / len(cnt), late_timeouts(agent.qnetwork_local)))
        env.close()
    print(f"~{row[0][1]:6.0f} episode ends per vector step: in front {row[0][0]:.4f} {row[2][0]:.4f} ms | under the act kernel {row[1][0]:.4f} {row[3][0]:.4f} ms"
          f" | difference {1e3 * ((row[1][0] + row[3][0]) - (row[0][0] + row[2][0])) / 2:+.1f} us | late-row timeouts {row[1][2] + row[3][2]}", flush=True)
