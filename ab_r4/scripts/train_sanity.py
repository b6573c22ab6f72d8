"""Sanity run: does the batched loop actually learn?  Trains a fresh IQN on the vector env for a few
thousand vector steps (one grad step per vector step) and evaluates on the reference's 30 eval worlds
before / during / after.  Not a benchmark."""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv

n_envs = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
with open(os.path.join(os.path.dirname(__file__), "..", "tests", "golden", "eval_config_seed3.json")) as f:
    cfg = json.load(f)
sched = dict(timesteps=[0, 1000000, 2000000], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10], min_start_goal_dis=[30.0, 35.0, 40.0])
total = n_envs * steps
env = VecMarineNavEnv(n_envs, seed=0, schedule=sched, timestep_scale=3_000_000 / total, device="cuda:0")  # whole curriculum over the run
eval_env = VecMarineNavEnv(30, device="cuda:0")
seed = int(sys.argv[3]) if len(sys.argv) > 3 else 100
agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=1_000_000, device="cuda:0", seed=seed, learning_starts=n_envs * 4,
                 target_update_interval=500, UPDATE_EVERY=1)
if os.environ.get("MN_ACT_TORCH_RNG") == "1":
    agent.use_library_rng = False
def ev(tag):
    r = agent.evaluation_vec(eval_env, cfg, greedy=True)
    print(f"[{tag}] eval: success {sum(r['successes'])}/30  mean return {np.mean(r['rewards']):.2f}", flush=True)
import io, contextlib
t0 = time.time()
obs = env.reset()
for it in range(steps):
    eps = agent.linear_eps(total)
    obs, *_ = agent.vec_step(env, obs, eps, 1.0, train_every=1, per_iter=n_envs)
    if it % (steps // 8) == 0:
        with contextlib.redirect_stdout(io.StringIO()) as buf:
            r = agent.evaluation_vec(eval_env, cfg, greedy=True)
        print(f"[step {it:5d} | env steps {agent.current_timestep:9d} | grad steps {agent.grad_steps:5d} | {time.time()-t0:5.1f}s] "
              f"eval success {sum(r['successes'])}/30  mean return {np.mean(r['rewards']):7.2f}", flush=True)
with contextlib.redirect_stdout(io.StringIO()):
    r = agent.evaluation_vec(eval_env, cfg, greedy=True)
print(f"[final | {time.time()-t0:5.1f}s] eval success {sum(r['successes'])}/30  mean return {np.mean(r['rewards']):7.2f}")
