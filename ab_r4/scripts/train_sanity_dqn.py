"""Sanity run for the DQN baseline learner (dqn/agent.py) on the vector env: train a fresh DQN (one update per vector
step) and evaluate the greedy policy on the reference's 30 evaluation worlds.  Not a benchmark.
python scripts/train_sanity_dqn.py [n_envs] [vector_steps]"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributional_rl_navigation_amd.dqn import DQNAgent
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 20000
with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "eval_config_seed3.json")) as f:
    cfg = list(json.load(f).values())
sched = dict(timesteps=[0, 1000000, 2000000], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10], min_start_goal_dis=[30.0, 35.0, 40.0])
total = n_envs * steps
env = VecMarineNavEnv(n_envs, seed=0, schedule=sched, timestep_scale=3_000_000 / total, device="cuda:0")
eval_env = VecMarineNavEnv(30, device="cuda:0", precision="f64")
agent = DQNAgent(device="cuda:0", buffer_size=1_000_000, batch_size=256, learning_starts=n_envs * 4, train_freq=1,
                 target_update_interval=500 * n_envs, seed=103)


@torch.no_grad()
def evaluate():
    obs = eval_env.load_worlds([VecMarineNavEnv.world_from_eval_config(c) for c in cfg]).clone()
    alive = torch.ones(30, dtype=torch.bool, device="cuda:0"); ret = torch.zeros(30, dtype=torch.float64, device="cuda:0")
    last = torch.zeros(30, dtype=torch.uint8, device="cuda:0")
    for t in range(1000):
        obs, r, d, info = eval_env.step(agent.policy.act_batch(obs))
        ret += torch.where(alive, (0.99 ** t) * r.double(), torch.zeros_like(ret))
        last = torch.where(alive, info, last)
        alive &= ~d.bool()
        if not bool(alive.any()):
            break
    return int((last == 4).sum()), float(ret.mean())


t0 = time.time()
every = max(1, steps // 8)
def cb(ag, it):
    if it % every == 0:
        s, r = evaluate()
        print(f"[step {it:6d} | env steps {ag.num_timesteps:10d} | updates {ag.n_updates:6d} | {time.time()-t0:5.1f}s] "
              f"eval success {s}/30  mean return {r:7.2f}", flush=True)
stats = agent.learn_vec(steps, env, callback=cb)
s, r = evaluate()
print(f"[final | {time.time()-t0:5.1f}s] eval success {s}/30  mean return {r:7.2f}   ({stats})")
