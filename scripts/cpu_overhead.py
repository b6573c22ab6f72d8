import sys, time, os
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
n = 65536
env = VecMarineNavEnv(n, seed=0, device="cuda:0")
env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device="cuda:0", seed=100, learning_starts=0)
agent.reset_under_act = os.environ.get("UNDER", "1") == "1"
obs = env.reset()
for _ in range(20):
    obs = agent.vec_step(env, obs, 1.0, 1.0, per_iter=n)[0]
torch.cuda.synchronize()
K = 400
c0 = time.process_time(); w0 = time.perf_counter()
for _ in range(K):
    obs = agent.vec_step(env, obs, 1.0, 1.0, per_iter=n)[0]
w1 = time.perf_counter(); c1 = time.process_time()
torch.cuda.synchronize()
w2 = time.perf_counter(); c2 = time.process_time()
print(f"python loop issue time {1e3*(w1-w0)/K:.3f} ms/step (wall, before sync); total wall {1e3*(w2-w0)/K:.3f} ms/step; "
      f"process CPU time {1e3*(c2-c0)/K:.3f} ms/step ({(c2-c0)/(w2-w0):.2f} cores busy)")
if len(sys.argv) > 1 and sys.argv[1] == "profile":
    import cProfile, pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(K):
        obs = agent.vec_step(env, obs, 1.0, 1.0, per_iter=n)[0]
    pr.disable(); torch.cuda.synchronize()
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
