"""Soak run: the full training loop (65 536 envs, act + step + replay + reset + train) for many vector steps;
checks finiteness, counters, replay consistency and that device memory does not grow."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.agent import IQNAgent
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
steps = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
grad_steps = int(sys.argv[2]) if len(sys.argv) > 2 else 1      # gradient steps per training event (16 = the cadence train_iqn runs)
update_every = int(sys.argv[3]) if len(sys.argv) > 3 else 4    # vector steps between training events
n = 65536
sched = dict(timesteps=[0, 1000000, 2000000], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10], min_start_goal_dis=[30.0, 35.0, 40.0])
env = VecMarineNavEnv(n, seed=0, schedule=sched, timestep_scale=3e6 / (n * steps) * n, device="cuda:0")
agent = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=100_000, device="cuda:0", seed=100, learning_starts=0, UPDATE_EVERY=update_every)
agent.grad_steps_per_update = grad_steps
agent.reset_under_act = True      # (the loop only looks at `obs` behind a device synchronisation)
obs = env.reset()
total = n * steps
t0 = time.time(); mem0 = None; done_total = 0
for it in range(steps):
    obs, r, d, info, loss = agent.vec_step(env, obs, agent.linear_eps(total), 1.0, per_iter=n)
    if it % 1000 == 999:
        torch.cuda.synchronize()
        mem = torch.cuda.memory_allocated()
        mem0 = mem0 or mem
        assert torch.isfinite(obs).all() and torch.isfinite(r).all(), it
        assert ((info != 0) == d.bool()).all()
        s, ep, tot = env.get_state(0, 4096)
        assert (tot == it + 1).all() and (ep >= 0).all() and (ep <= 1001).all()
        w = env.get_worlds(0, 256)
        print(f"step {it+1:6d}  {n*(it+1)/(time.time()-t0)/1e6:6.1f} M env steps/s  mem {mem/1e6:.0f} MB  done this step {int(d.sum())}  "
              f"grad steps {agent.grad_steps}  loss {float(loss) if loss is not None else float('nan'):.3f}  "
              f"world sizes {sorted(set((x['n_cores'], x['n_obs']) for x in w))[-1]}", flush=True)
        assert mem <= mem0 * 1.05
        assert loss is None or torch.isfinite(loss).all()
        if getattr(agent, "_fused", None) is not None:      # the one-launch gradient step's placement diagnostic
            assert agent._fused.xcd_misplaced() == 0
print("soak ok")
