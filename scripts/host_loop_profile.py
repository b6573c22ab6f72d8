import cProfile, pstats, sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
n = 4096
env = VecMarineNavEnv(n, seed=0, device="cuda:0", precision="f64")
agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device="cuda:0", seed=1, learning_starts=0, UPDATE_EVERY=1)
agent.grad_steps_per_update = 1
agent.learn_vec(total_vector_steps=500, train_env=env, verbose=False)
torch.cuda.synchronize()
t0 = time.time()
agent.learn_vec(total_vector_steps=5000, train_env=env, verbose=False)
torch.cuda.synchronize()
print("ms per vector step (4096 envs, 1 grad step):", (time.time() - t0) / 5000 * 1e3)
# GPU-only time of the same: events around 200 steps enqueued ... (host-bound if wall >> this)
pr = cProfile.Profile(); pr.enable()
agent.learn_vec(total_vector_steps=3000, train_env=env, verbose=False)
torch.cuda.synchronize(); pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
