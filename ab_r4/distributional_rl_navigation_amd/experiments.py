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
from .planners import planner_act_batch

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


def _episode_record(world, params, name, actions, traj, cvars=None, quantiles=None, taus=None, seed=15):
    """One `ep_data` entry of the reference's exp_data JSON: MarineNavEnv.episode_data() (marinenav_env.py:557-622) of
    the finished episode plus, for the IQN policies, robot.actions_cvars / actions_quantiles / actions_taus
    (run_experiments.py:62-69)."""
    p = params
    ep = {"env": {}, "robot": {}}
    e = ep["env"]
    e["seed"] = seed
    e["width"], e["height"], e["r"], e["v_rel_max"], e["p"] = p.width, p.height, p.core_r, p.v_rel_max, p.p
    e["v_range"] = [p.v_range[0], p.v_range[1]]; e["obs_r_range"] = [p.obs_r_range[0], p.obs_r_range[1]]
    e["clear_r"] = p.clear_r
    e["start"] = [float(v) for v in world["start"]]; e["goal"] = [float(v) for v in world["goal"]]
    e["goal_dis"], e["timestep_penalty"], e["collision_penalty"] = p.goal_dis, p.timestep_penalty, p.collision_penalty
    e["goal_reward"], e["discount"] = p.goal_reward, p.discount
    c, o = world["cores"], world["obstacles"]
    e["cores"] = {"positions": [[float(r[0]), float(r[1])] for r in c], "clockwise": [int(r[2]) for r in c],
                  "Gamma": [float(r[3]) for r in c]}
    e["obstacles"] = {"positions": [[float(r[0]), float(r[1])] for r in o], "r": [float(r[2]) for r in o]}
    ep["robot"] = {"dt": p.dt, "N": p.N, "length": 1.0, "width": 0.5, "r": p.robot_r, "max_speed": p.max_speed,
                   "a": [p.a[0], p.a[1], p.a[2]], "w": [p.w[0], p.w[1], p.w[2]],
                   "init_theta": float(world["init_theta"]), "init_speed": float(world["init_speed"]),
                   "sonar": {"range": p.sonar_range, "angle": p.sonar_angle, "num_beams": p.num_beams},
                   "action_history": [int(a) for a in actions], "trajectory": [[float(q[0]), float(q[1])] for q in traj]}
    if cvars is not None:
        ep["robot"]["actions_cvars"] = [float(v) for v in cvars]
        ep["robot"]["actions_quantiles"] = [q.tolist() for q in quantiles]       # each [1][32][9], as act_eval returns
        ep["robot"]["actions_taus"] = [t.tolist() for t in taus]                 # each [1][32][1]
    return ep


@torch.no_grad()
def _classical_episodes(worlds, name, device, max_steps, seed):
    """All of `worlds` under the classical baseline `name` ("APF" / "BA"), one episode each, as ONE launch: the policy runs inside the
    rollout kernel (VecMarineNavEnv.rollout_policy -> mn_rollout_policy).  Same result record as the launch-per-step path."""
    num = len(worlds)
    env = VecMarineNavEnv(num, device=device, precision="f64")
    _configure(env)
    env.load_worlds(worlds)
    dev = env.device
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    tr = env.rollout_policy(max_steps, name)
    e1.record()
    reward, done, info, acts = tr["reward"], tr["done"].bool(), tr["info"], tr["action"]
    T = reward.shape[0]
    a_tab = torch.tensor(env.params.a[:], device=dev); w_tab = torch.tensor(env.params.w[:], device=dev)
    energy_tab = ((a_tab / a_tab.max()).abs().view(3, 1) + (w_tab / w_tab.max()).abs().view(1, 3)).reshape(-1)
    alive = torch.ones(num, dtype=torch.bool, device=dev)
    ret = torch.zeros(num, dtype=torch.float64, device=dev); energy = torch.zeros_like(ret)
    length = torch.zeros(num, dtype=torch.int64, device=dev)
    last_info = torch.zeros(num, dtype=torch.uint8, device=dev)
    for t in range(T):      # the bookkeeping of the per-step loop below, on the traces (same operations, same order)
        ret += torch.where(alive, (env.discount ** t) * reward[t].double(), torch.zeros_like(ret))
        length += alive.long()
        energy += torch.where(alive, energy_tab[acts[t].clamp_min(0).long()].double(), torch.zeros_like(energy))
        last_info = torch.where(alive, info[t], last_info)
        alive = alive & ~done[t]
        if not bool(alive.any()):
            break
    torch.cuda.synchronize(dev)
    length_h, info_h, acts_h = length.cpu().numpy(), last_info.cpu().numpy(), acts.cpu().numpy()
    dtN = env.params.dt * env.params.N
    per_action = e0.elapsed_time(e1) * 1e-3 / max(1, int(length_h.sum()))
    rec = dict(success=[bool(v) for v in info_h == 4], out_of_area=[bool(v) for v in info_h == 1],
               time=[float(dtN * l) for l in length_h], energy=[float(v) for v in energy.cpu().numpy()],
               reward=[float(v) for v in ret.cpu().numpy()],
               actions=[[int(x) for x in acts_h[:length_h[i], i]] for i in range(num)],
               computation_times=[per_action] * int(length_h.sum()))
    env.close()
    return rec


def run_experiment(agent, n_obs, n_cores, num=500, seed=15, policies=POLICIES, device="cuda:0", max_steps=1000, dqn=None,
                   capture=False, classical_rollout=True):
    """run_experiments.py:213-282 for the IQN policies, the classical APF / BA baselines and (when `dqn`, a
    `dqn.DQNPolicy`, is given and "DQN" is in `policies`) the greedy DQN baseline.  Returns {policy: dict(success, time, energy,
    out_of_area, reward, actions)} with one entry per world.  With `capture` each policy also gets the reference's `ep_data`
    list (run_experiments.py:26-69,262-282): per episode the episode_data() dict incl. the sub-step trajectory, and for the
    IQN policies the per-action CVaR level, quantile values [1,32,9] and taus [1,32,1] of IQNAgent.act_eval -- the whole
    `exp_data` JSON the reference dumps.  `computation_times` (run_experiments.py:30,37-44,254-255: the wall-clock seconds of every
    act call, flattened over a policy's episodes) is the batched equivalent: the device time of the step's act launch(es) for that
    policy group (HIP events) divided by the rows the launch served, one entry per step of every episode -- the amortised cost of one
    action, which is what `avg_compute_t` (run_experiments.py:274) averages."""
    worlds = generate_worlds(num, n_obs, n_cores, seed, device)
    # APF / BA: the policy is a device function inside the episode rollout kernel -- one launch per policy for all worlds -- unless the
    # per-sub-step trajectory is wanted (`capture`), which the launch-per-step path below records
    requested = tuple(policies)
    rolled = {}
    if classical_rollout and not capture:
        rolled = {name: _classical_episodes(worlds, name, device, max_steps, seed) for name in requested if name in ("APF", "BA")}
        policies = tuple(p for p in requested if p not in rolled)
        if not policies:
            return {name: rolled[name] for name in requested}, worlds
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
    cap_cv, cap_q, cap_t, cap_traj = [], [], [], []
    act_events = {}                                                # group -> [(start, end)] per step; groups: "IQN" (one launch for all IQN policies), planners
    alive_hist = []                                                # per step: live envs per policy (before the step)

    def timed(group, fn):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = fn()
        e1.record()
        act_events.setdefault(group, []).append((e0, e1))
        return r
    if capture:
        env.enable_trajectory()
    if agent is not None:
        agent.qnetwork_local.eval()
    for t in range(max_steps):
        a = torch.zeros(n, dtype=torch.int32, device=dev)
        alive_hist.append(alive.view(len(policies), num).sum(dim=1))
        if iqn_idx.numel():
            o = obs[iqn_idx].contiguous()

            def iqn_act():      # what the reference times (run_experiments.py:35-44): adjust_cvar + act_eval / act
                cv_ = torch.where(adaptive[iqn_idx], agent.adjust_cvar_batch(o), fixed[iqn_idx])   # agent.py:249-267 per row
                if capture:      # act_eval / act_adaptive_eval (agent.py:217-247): the action AND what it was chosen from
                    return (cv_,) + tuple(agent.act_eval_batch(o, 0.0, cv_))
                return cv_, agent.act_batch(o, 0.0, cv_), None, None
            cv, a_iqn, quant, taus = timed("IQN", iqn_act)
            a[iqn_idx] = a_iqn
            if capture:
                cap_cv.append(cv.cpu().numpy()); cap_q.append(quant.cpu().numpy()); cap_t.append(taus.cpu().numpy())
        for name, rows in classical.items():                                 # APF.py:17-78 / BA.py:14-72
            if name == "DQN":                                                # run_experiments.py:86 (greedy predict)
                a[rows] = timed(name, lambda: dqn.act_batch(obs[rows]))
                continue
            a[rows] = timed(name, lambda: planner_act_batch(obs[rows], name, env.params.a[:], env.params.w[:]))      # one HIP launch (mn_planner_act)
        obs, reward, done, info = env.step(a)
        if capture:
            cap_traj.append(env.get_trajectory())
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
    torch.cuda.synchronize(dev)
    group_rows = {"IQN": max(1, int(iqn_idx.numel()))}
    step_s = {g: [e0.elapsed_time(e1) * 1e-3 / group_rows.get(g, num) for e0, e1 in evs] for g, evs in act_events.items()}
    alive_h = torch.stack(alive_hist).cpu().numpy() if alive_hist else np.zeros((0, len(policies)), dtype=np.int64)
    out = {}
    for p, name in enumerate(policies):
        sl = slice(p * num, (p + 1) * num)
        out[name] = dict(success=[bool(v) for v in info_h[sl] == 4], out_of_area=[bool(v) for v in info_h[sl] == 1],
                         time=[float(dtN * l) for l in length_h[sl]], energy=[float(v) for v in energy_h[sl]],
                         reward=[float(v) for v in ret_h[sl]],
                         actions=[[int(x) for x in acts_h[:length_h[i], i]] for i in range(p * num, (p + 1) * num)])
        ts = step_s.get(name if name in classical else "IQN", [])
        # one entry per act call of the reference = per live episode and step: the step's amortised per-row device time
        out[name]["computation_times"] = [float(ts[t_]) for t_ in range(len(ts)) for _ in range(int(alive_h[t_, p]))]
        if capture:
            iqn_pos = {int(g_): k for k, g_ in enumerate(iqn_idx.cpu().numpy())}     # env row -> row of the IQN captures
            eps_ = []
            for i in range(p * num, (p + 1) * num):
                L = int(length_h[i])
                traj = [q for t_ in range(L) for q in cap_traj[t_][i]]
                if i in iqn_pos:
                    k = iqn_pos[i]
                    eps_.append(_episode_record(worlds[i - p * num], env.params, name, acts_h[:L, i], traj,
                                                cvars=[cap_cv[t_][k] for t_ in range(L)],
                                                quantiles=[cap_q[t_][k:k + 1] for t_ in range(L)],
                                                taus=[cap_t[t_][k:k + 1] for t_ in range(L)], seed=seed))
                else:
                    eps_.append(_episode_record(worlds[i - p * num], env.params, name, acts_h[:L, i], traj, seed=seed))
            out[name]["ep_data"] = eps_
    env.close()
    out.update(rolled)
    return {name: out[name] for name in requested}, worlds
