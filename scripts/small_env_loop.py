import sys, time, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
env = VecMarineNavEnv(n, seed=0, device="cuda:0", precision="f64")
agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device="cuda:0", seed=1, learning_starts=0, UPDATE_EVERY=1)
agent.grad_steps_per_update = max(1, round(n / 65536 * 16))
agent.learn_vec(total_vector_steps=3000, train_env=env, verbose=False)
torch.cuda.synchronize()
