"""Batched counterpart of the IQN part of the reference's run_experiments.py (:19-72,192-282).

The reference runs 500 randomised worlds x {adaptive IQN, IQN cvar 0.25 / 0.5 / 0.75 / 1.0} one episode at a
time on the CPU (`exp_setup_5`: fixed start (5,5) / goal (45,45), `set_boundary = True`, `robot.N = 5`,
`random_reset_state = False`, every test env seeded with 15 so that all agents see the same world sequence).
Here the worlds are generated once from the same RNG stream (bit-identical to the reference's), replicated
per policy into ONE vector env and all 500 x 5 episodes are stepped side by side on the GPU.
"""
import numpy as np
import torch

from .marinenav_env.vec_env import VecMarineNavEnv
from .planners import apf_act_batch, ba_act_batch

POLICIES = ("adaptive_IQN", "IQN_0.25", "IQN_0.5", "IQN_0.75", "IQN_1.0", "APF", "BA")   # run_experiments.py:216 (minus DQN)
ALL_POLICIES = POLICIES[:5] + ("DQN",) + POLICIES[5:]                                        # run_experiments.py:216, needs `dqn=`
_CVAR = {"IQN_0.25": 0.25, "IQN_0.5": 0.5, "IQN_0.75": 0.75, "IQN_1.0": 1.0}


def _configure(env):
    """exp_setup_5 (run_experiments.py:192-211)."""
    env.set_attrs(reset_start_and_goal=False, random_reset_state=False, set_boundary=True, obs_r_range=[1, 3], N=5)
    env.set_start_goal([5.0, 5.0], [45.0, 45.0])


def generate_worlds(num, n_obs, n_cores, seed=15, device="cuda:0"):
    """The world sequence every test env of the reference sees: `num` consecutive reset()s of one
    RandomState(seed) stream under exp_setup_5 settings."""
    gen = VecMarineNavEnv(1, seeds=[seed], device=device, precision="f64")
    _configure(gen)
    gen.set_attrs(num_cores=n_cores, num_obs=n_obs)
    worlds = []
    for _ in range(num):
        gen.reset()
        worlds.append(gen.get_worlds(0, 1)[0])
    gen.close()
    return worlds


@torch.no_grad()
def run_experiment(agent, n_obs, n_cores, num=500, seed=15, policies=POLICIES, device="cuda:0", max_steps=1000, dqn=None):
    """run_experiments.py:213-282 for the IQN policies, the classical APF / BA baselines and (when `dqn`, a
    `dqn.DQNPolicy`, is given and "DQN" is in `policies`) the greedy DQN baseline.  Returns {policy: dict(success, time, energy,
    out_of_area, reward, actions)} with one entry per world (the reference's exp_data schema minus the
    per-step quantile dumps and wall-clock timings)."""
    worlds = generate_worlds(num, n_obs, n_cores, seed, device)
    n = num * len(policies)
    env = VecMarineNavEnv(n, device=device, precision="f64")
    _configure(env)
    obs = env.load_worlds(worlds * len(policies)).clone()          # policy p owns envs [p*num, (p+1)*num)
    dev = env.device
    fixed = torch.ones(n, device=dev)
    adaptive = torch.zeros(n, dtype=torch.bool, device=dev)
    classical = {}                                                 # policy name -> env rows driven by a planner
    iqn_rows = torch.zeros(n, dtype=torch.bool, device=dev)
    for p, name in enumerate(policies):
        rows = slice(p * num, (p + 1) * num)
        if name in ("APF", "BA", "DQN"):
            if name == "DQN" and dqn is None:
                raise ValueError("policy 'DQN' needs run_experiment(..., dqn=DQNPolicy.load(...))")
            classical[name] = rows
            continue
        iqn_rows[rows] = True
        if name == "adaptive_IQN":
            adaptive[rows] = True
        else:
            fixed[rows] = _CVAR[name]
    iqn_idx = torch.nonzero(iqn_rows).view(-1)
    a_tab = torch.tensor(env.params.a[:], device=dev); w_tab = torch.tensor(env.params.w[:], device=dev)
    energy_tab = ((a_tab / a_tab.max()).abs().view(3, 1) + (w_tab / w_tab.max()).abs().view(1, 3)).reshape(-1)
    alive = torch.ones(n, dtype=torch.bool, device=dev)
    ret = torch.zeros(n, dtype=torch.float64, device=dev); energy = torch.zeros_like(ret)
    length = torch.zeros(n, dtype=torch.int64, device=dev)
    last_info = torch.zeros(n, dtype=torch.uint8, device=dev)
    acts = torch.full((max_steps, n), -1, dtype=torch.int32, device=dev)
    if agent is not None:
        agent.qnetwork_local.eval()
    for t in range(max_steps):
        a = torch.zeros(n, dtype=torch.int32, device=dev)
        if iqn_idx.numel():
            o = obs[iqn_idx]
            cv = torch.where(adaptive[iqn_idx], agent.adjust_cvar_batch(o), fixed[iqn_idx])   # agent.py:249-267 per row
            a[iqn_idx] = agent.act_batch(o, 0.0, cv)
        for name, rows in classical.items():                                 # APF.py:17-78 / BA.py:14-72
            if name == "DQN":                                                # run_experiments.py:86 (greedy predict)
                a[rows] = dqn.act_batch(obs[rows])
                continue
            fn = apf_act_batch if name == "APF" else ba_act_batch
            a[rows] = fn(obs[rows].double(), a_tab.double(), w_tab.double()).to(torch.int32)
        obs, reward, done, info = env.step(a)
        ret += torch.where(alive, (env.discount ** t) * reward.double(), torch.zeros_like(ret))
        length += alive.long()
        energy += torch.where(alive, energy_tab[a.long()].double(), torch.zeros_like(energy))
        acts[t] = torch.where(alive, a, torch.full_like(a, -1))
        last_info = torch.where(alive, info, last_info)
        alive = alive & ~done.bool()
        if not bool(alive.any()):
            break
    if agent is not None:
        agent.qnetwork_local.train()
    length_h = length.cpu().numpy(); info_h = last_info.cpu().numpy(); acts_h = acts.cpu().numpy()
    ret_h = ret.cpu().numpy(); energy_h = energy.cpu().numpy()
    dtN = env.params.dt * env.params.N
    out = {}
    for p, name in enumerate(policies):
        sl = slice(p * num, (p + 1) * num)
        out[name] = dict(success=[bool(v) for v in info_h[sl] == 4], out_of_area=[bool(v) for v in info_h[sl] == 1],
                         time=[float(dtN * l) for l in length_h[sl]], energy=[float(v) for v in energy_h[sl]],
                         reward=[float(v) for v in ret_h[sl]],
                         actions=[[int(x) for x in acts_h[:length_h[i], i]] for i in range(p * num, (p + 1) * num)])
    env.close()
    return out, worlds
