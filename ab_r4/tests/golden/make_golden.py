#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by IMPORTING the Python reference.

Runs only in the build container (needs /root/reference).  Nothing here is used at
test/bench time: the tests read the emitted .npz/.json data files only.

The reference needs two shims to import under numpy 2.x without gym installed
(SURVEY.md Appendix B): a stub ``gym`` module and ``np.infty``.

Outputs (all small, committed):
  g1_reset.npz          world generation: seeds x world sizes x 3 consecutive resets
  g2_trace_*.npz        1000-step traces with caller-side reset on done
  g3_single_step.npz    2048 independent (world, state, action) -> step outputs
  g4_sonar_edge.npz     hand-built sonar/observation edge cases
  g5_velocity.npz       current-field samples (inside core, far field, 0/1-core worlds)
  g6_pretrained_replay.npz   stored action sequences of the reference's own evaluation
                        files + the reference's stored returns/success/time/energy
  eval_config_seed3.json     data file shipped by the reference (30 evaluation worlds)
  g7_iqn.npz            IQN forward / loss / grads with injected taus, adjust_cvar, linear_eps
  g8_boundary_trace.npz set_boundary = True, robot.N = 5 trace (run_experiments.py settings)
  g9_planners.npz       APF / BA baseline actions for 2304 observations
  pretrained_IQN_seed3/ checkpoint data files shipped by the reference (weights only)
  g12_replay.npz        the reference ReplayBuffer: 1500 adds into maxlen 1000, contents, one sample()
  g15_replay_nstep.npz  the reference ReplayBuffer with n_step = 3: 40 adds into maxlen 25, contents
  g13_learn_loop.npz    bookkeeping of the reference IQNAgent.learn loop over 400 timesteps on the reference env
  g14_iqn_episodes.npz  run_experiments.py's evaluation_IQN loop on the reference env + pretrained agent, injected taus:
                        per-step action / CVaR / quantiles / taus, per-episode outcome and trajectory
  (g10 / g11: make_golden_dqn.py)
"""
import contextlib
import io
import json
import os
import shutil
import sys
import types
import warnings

import numpy as np

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))

warnings.filterwarnings("ignore", category=PendingDeprecationWarning)
if not hasattr(np, "infty"):
    np.infty = np.inf


def _install_gym_stub():
    gym = types.ModuleType("gym")

    class Env:
        def close(self):
            pass

    class Discrete:
        def __init__(self, n):
            self.n = n

    class Box:
        def __init__(self, low, high, dtype=None):
            self.low, self.high, self.dtype = low, high, dtype

    spaces = types.ModuleType("gym.spaces")
    spaces.Discrete, spaces.Box = Discrete, Box
    envs = types.ModuleType("gym.envs")
    reg = types.ModuleType("gym.envs.registration")
    reg.register = lambda **kw: None
    envs.registration = reg
    gym.Env, gym.spaces, gym.envs = Env, spaces, envs
    sys.modules.update({"gym": gym, "gym.spaces": spaces, "gym.envs": envs,
                        "gym.envs.registration": reg})


_install_gym_stub()
sys.path.insert(0, REF)
from marinenav_env.envs.marinenav_env import MarineNavEnv, Core, Obstacle  # noqa: E402

INFO_CODE = {"normal": 0, "out of boundary": 1, "too long episode": 2, "collision": 3, "reach goal": 4}
MAXC, MAXO = 8, 10


def quiet(fn, *a, **kw):
    with contextlib.redirect_stdout(io.StringIO()):
        return fn(*a, **kw)


def world_arrays(env):
    c = np.zeros((MAXC, 4))
    o = np.zeros((MAXO, 3))
    for i, k in enumerate(env.cores):
        c[i] = [k.x, k.y, float(k.clockwise), k.Gamma]
    for i, k in enumerate(env.obstacles):
        o[i] = [k.x, k.y, k.r]
    return c, o, len(env.cores), len(env.obstacles)


def robot_state(env):
    r = env.robot
    return np.array([r.x, r.y, r.theta, r.speed, r.velocity[0], r.velocity[1]])


def peek_next_double(env):
    st = env.rd.get_state()
    v = env.rd.random_sample()
    env.rd.set_state(st)
    return v


WORLD_SIZES = [(4, 6, 30.0), (6, 8, 35.0), (8, 10, 40.0), (8, 5, 25.0)]


def g1_reset():
    rec = {k: [] for k in ("seed", "size", "start", "goal", "cores", "obs", "ncores", "nobs",
                           "theta0", "speed0", "obs0", "state0", "next_double")}
    for seed in (0, 1, 2, 3, 348):
        for (nc, no, md) in WORLD_SIZES:
            env = MarineNavEnv(seed=seed)
            env.num_cores, env.num_obs, env.min_start_goal_dis = nc, no, md
            for _ in range(3):
                ob = env.reset()
                c, o, n1, n2 = world_arrays(env)
                rec["seed"].append(seed)
                rec["size"].append([nc, no, md])
                rec["start"].append(env.start.copy())
                rec["goal"].append(env.goal.copy())
                rec["cores"].append(c)
                rec["obs"].append(o)
                rec["ncores"].append(n1)
                rec["nobs"].append(n2)
                rec["theta0"].append(env.robot.init_theta)
                rec["speed0"].append(env.robot.init_speed)
                rec["obs0"].append(ob)
                rec["state0"].append(robot_state(env))
                rec["next_double"].append(peek_next_double(env))
    np.savez_compressed(os.path.join(OUT, "g1_reset.npz"), **{k: np.array(v) for k, v in rec.items()})


def g1b_eval_worlds():
    """create_eval_configs semantics (train_IQN_model.py:123-148): seed 348, fixed start/goal."""
    env = MarineNavEnv(seed=348)
    env.obs_r_range = [1, 3]
    env.reset_start_and_goal = False
    env.start = np.array([5.0, 5.0])
    env.goal = np.array([45.0, 45.0])
    # the regenerated worlds are compared with eval_config_seed3.json by the tests
    shutil.copyfile(os.path.join(REF, "pretrained_models/IQN/seed_3/eval_config.json"),
                    os.path.join(OUT, "eval_config_seed3.json"))


def trace(seed, schedule, name, n_steps=1000, world=None):
    env = MarineNavEnv(seed=seed, schedule=schedule)
    if world is not None:
        env.num_cores, env.num_obs, env.min_start_goal_dis = world
    ar = np.random.RandomState(seed + 1000)
    actions = ar.randint(9, size=n_steps)
    obs0 = quiet(env.reset)
    rec = {k: [] for k in ("obs", "reward", "done", "info", "state", "ep_t", "tot_t", "reset_obs")}
    worlds = [world_arrays(env) + (env.start.copy(), env.goal.copy())]
    for t in range(n_steps):
        ob, r, d, info = env.step(int(actions[t]))
        rec["obs"].append(ob)
        rec["reward"].append(r)
        rec["done"].append(d)
        rec["info"].append(INFO_CODE[info["state"]])
        rec["state"].append(robot_state(env))
        rec["ep_t"].append(env.episode_timesteps)
        rec["tot_t"].append(env.total_timesteps)
        if d:
            ro = quiet(env.reset)
            rec["reset_obs"].append(ro)
            worlds.append(world_arrays(env) + (env.start.copy(), env.goal.copy()))
        else:
            rec["reset_obs"].append(np.zeros(26))
    out = {k: np.array(v) for k, v in rec.items()}
    out["actions"] = actions
    out["obs0"] = obs0
    out["seed"] = seed
    out["world_cores"] = np.array([w[0] for w in worlds])
    out["world_obs"] = np.array([w[1] for w in worlds])
    out["world_n"] = np.array([[w[2], w[3]] for w in worlds])
    out["world_start"] = np.array([w[4] for w in worlds])
    out["world_goal"] = np.array([w[5] for w in worlds])
    out["size"] = np.array([env.num_cores, env.num_obs, env.min_start_goal_dis])
    if schedule is not None:
        out["sched_timesteps"] = np.array(schedule["timesteps"])
        out["sched_num_cores"] = np.array(schedule["num_cores"])
        out["sched_num_obstacles"] = np.array(schedule["num_obstacles"])
        out["sched_min_dis"] = np.array(schedule["min_start_goal_dis"])
    np.savez_compressed(os.path.join(OUT, name), **out)


def g2_traces():
    trace(0, None, "g2_trace_seed0_default.npz")
    trace(1, None, "g2_trace_seed1_stage0.npz", world=(4, 6, 30.0))
    trace(2, None, "g2_trace_seed2_stage2.npz", world=(8, 10, 40.0))
    # a compressed curriculum so the schedule lookup changes inside the trace
    # (episodes of a random policy usually time out at step 1001, so resets land near 1001, 2002, 3003)
    sched = dict(timesteps=[0, 900, 2100], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10],
                 min_start_goal_dis=[30.0, 35.0, 40.0])
    trace(5, sched, "g2_trace_seed5_schedule.npz", n_steps=3100)


def g8_boundary_trace():
    """run_experiments.py:192-211 style settings: set_boundary = True, robot.N = 5, fixed start/goal near
    the map edge so that 'out of boundary' terminations occur; caller-side reset on done."""
    env = MarineNavEnv(seed=21)
    env.set_boundary = True
    env.robot.N = 5
    env.reset_start_and_goal = False
    env.start = np.array([3.0, 4.0])
    env.goal = np.array([46.0, 45.0])
    env.num_cores, env.num_obs = 8, 8
    ar = np.random.RandomState(77)
    n_steps = 1500
    actions = ar.randint(9, size=n_steps)
    obs0 = env.reset()
    rec = {k: [] for k in ("obs", "reward", "done", "info", "state", "ep_t", "reset_obs")}
    for t in range(n_steps):
        ob, r, d, info = env.step(int(actions[t]))
        rec["obs"].append(ob); rec["reward"].append(r); rec["done"].append(d)
        rec["info"].append(INFO_CODE[info["state"]]); rec["state"].append(robot_state(env)); rec["ep_t"].append(env.episode_timesteps)
        rec["reset_obs"].append(env.reset() if d else np.zeros(26))
    out = {k: np.array(v) for k, v in rec.items()}
    out.update(actions=actions, obs0=obs0, seed=21, start=env.start, goal=env.goal)
    np.savez_compressed(os.path.join(OUT, "g8_boundary_trace.npz"), **out)


def g9_planners():
    """Classical baselines (APF.py:17-78, BA.py:14-155) on the 2048 observations of g3 plus synthetic
    corner cases: expected action indices."""
    import APF, BA
    z = np.load(os.path.join(OUT, "g3_single_step.npz"))
    obs = z["obs"].copy()
    rng = np.random.RandomState(99)
    extra = obs[rng.randint(len(obs), size=256)].copy()
    extra[:64, :2] *= 1e-4                      # near-zero velocity branches
    extra[64:128, 4:] = 0.0                     # no sonar returns
    for i in range(128, 192):                   # exactly one / two returns
        pts = extra[i, 4:].reshape(11, 2); keep = rng.choice(11, size=1 + (i % 2), replace=False)
        m = np.zeros(11, bool); m[keep] = True
        pts[~m] = 0.0
        pts[m] = rng.uniform(-8, 8, size=(m.sum(), 2))
    for i in range(192, 256):                   # vertical wall: same x for all returns
        pts = extra[i, 4:].reshape(11, 2); pts[:, 0] = rng.uniform(1, 8); pts[:, 1] = rng.uniform(-6, 6, size=11)
    obs = np.concatenate([obs, extra])
    env = MarineNavEnv(seed=0)
    apf = APF.APF_agent(env.robot.a, env.robot.w)
    ba = BA.BA_agent(env.robot.a, env.robot.w)
    a_apf = np.array([apf.act(o) for o in obs]); a_ba = np.array([ba.act(o) for o in obs])
    np.savez_compressed(os.path.join(OUT, "g9_planners.npz"), obs=obs, apf=a_apf, ba=a_ba, a=env.robot.a, w=env.robot.w)


def g3_single_step(n_worlds=64, per_world=32):
    rng = np.random.RandomState(777)
    rec = {k: [] for k in ("cores", "obs_tab", "n", "start", "goal", "state_in", "action", "ep_t",
                           "obs", "reward", "done", "info", "state_out")}
    for wi in range(n_worlds):
        nc, no, md = WORLD_SIZES[wi % 4]
        if wi % 16 == 15:
            nc = 0            # no vortices
        if wi % 16 == 14:
            nc = 1            # scalar-idx KDTree path (marinenav_env.py:428-429)
        env = MarineNavEnv(seed=10_000 + wi)
        env.num_cores, env.num_obs, env.min_start_goal_dis = nc, no, md
        env.reset()
        c, o, n1, n2 = world_arrays(env)
        for j in range(per_world):
            # random state anywhere in the map; a third of them near an obstacle or the goal so
            # that collision / goal / sonar branches are exercised
            mode = j % 3
            if mode == 1 and n2 > 0:
                k = env.obstacles[rng.randint(n2)]
                ang = rng.uniform(0, 2 * np.pi)
                dist = k.r + rng.uniform(0.3, 6.0)
                x, y = k.x + dist * np.cos(ang), k.y + dist * np.sin(ang)
            elif mode == 2:
                ang = rng.uniform(0, 2 * np.pi)
                dist = rng.uniform(0.5, 6.0)
                x, y = env.goal[0] + dist * np.cos(ang), env.goal[1] + dist * np.sin(ang)
            else:
                x, y = rng.uniform(0, 50, size=2)
            theta = rng.uniform(0, 2 * np.pi)
            speed = rng.uniform(0, 2.0)
            a = int(rng.randint(9))
            ep_t = int(rng.choice([0, 5, 999, 1000]))
            env.robot.x, env.robot.y, env.robot.theta, env.robot.speed = float(x), float(y), float(theta), float(speed)
            env.robot.velocity = np.zeros(2)
            env.episode_timesteps = ep_t
            sin = np.array([x, y, theta, speed])
            ob, r, d, info = env.step(a)
            rec["cores"].append(c); rec["obs_tab"].append(o); rec["n"].append([n1, n2])
            rec["start"].append(env.start.copy()); rec["goal"].append(env.goal.copy())
            rec["state_in"].append(sin); rec["action"].append(a); rec["ep_t"].append(ep_t)
            rec["obs"].append(ob); rec["reward"].append(r); rec["done"].append(d)
            rec["info"].append(INFO_CODE[info["state"]]); rec["state_out"].append(robot_state(env))
    np.savez_compressed(os.path.join(OUT, "g3_single_step.npz"), **{k: np.array(v) for k, v in rec.items()})


def g4_sonar_edge():
    """Observation-only cases: world + robot pose -> get_observation()."""
    cases = []

    def add(name, obstacles, x, y, theta, goal=(45.0, 45.0), vel=(0.3, -0.2)):
        env = MarineNavEnv(seed=0)
        env.cores.clear()
        env.obstacles.clear()
        for (ox, oy, r) in obstacles:
            env.obstacles.append(Obstacle(ox, oy, r))
        env.goal = np.array(goal)
        env.robot.x, env.robot.y, env.robot.theta, env.robot.speed = x, y, theta, 1.0
        env.robot.velocity = np.array(vel)
        ob = env.get_observation()
        o = np.zeros((MAXO, 3))
        for i, ob_ in enumerate(obstacles):
            o[i] = ob_
        cases.append(dict(name=name, obs_tab=o, n_obs=len(obstacles), pose=[x, y, theta],
                          goal=list(goal), vel=list(vel), obs=ob))

    phi = (2 * np.pi / 3) / 10
    # beam i exactly vertical / within and outside the 1e-3 snap window
    for i in (0, 5, 10):
        rel = -np.pi / 3 + i * phi
        for vert in (np.pi / 2, 3 * np.pi / 2):
            for eps in (0.0, 5e-4, -9.9e-4, 1.5e-3, -1.01e-3):
                th = vert - rel + eps
                while th < 0:
                    th += 2 * np.pi
                while th >= 2 * np.pi:
                    th -= 2 * np.pi
                sign = 1.0 if vert < np.pi else -1.0
                add(f"vert_b{i}_{vert:.2f}_{eps}", [(20.0, 20.0 + sign * 6.0, 2.0), (21.5, 20.0 + sign * 4.0, 1.0)],
                    20.0, 20.0, th)
    # robot inside a circle (nearer wall behind -> no hit; nearer wall ahead -> hit)
    add("inside_behind", [(20.5, 20.0, 3.0)], 20.0, 20.0, np.pi)       # nearer wall is at -x side? heading -x
    add("inside_ahead", [(20.5, 20.0, 3.0)], 20.0, 20.0, 0.0)
    add("inside_center_off", [(20.0, 21.0, 2.5)], 20.0, 20.0, np.pi / 2 - 0.2)
    # break-quirk: list order matters (near obstacle listed after a far one and vice versa)
    near, far, mid = (24.0, 20.0, 1.0), (28.0, 20.0, 2.0), (26.0, 20.3, 0.8)
    for order_name, order in (("nfm", [near, far, mid]), ("fnm", [far, near, mid]), ("fmn", [far, mid, near]),
                              ("mfn", [mid, far, near]), ("nmf", [near, mid, far]), ("mnf", [mid, near, far])):
        add("break_" + order_name, order, 20.0, 20.0, 0.0)
        add("break2_" + order_name, order, 20.0, 20.0, 0.15)
    # tangent / range boundary
    add("tangent", [(25.0, 21.0, 1.0)], 20.0, 20.0, 0.0)
    add("tangent_eps_in", [(25.0, 20.999999, 1.0)], 20.0, 20.0, 0.0)
    add("tangent_eps_out", [(25.0, 21.000001, 1.0)], 20.0, 20.0, 0.0)
    add("range_in", [(31.0, 20.0, 1.0000001)], 20.0, 20.0, 0.0)
    add("range_out", [(31.0, 20.0, 0.9999999)], 20.0, 20.0, 0.0)
    add("behind", [(15.0, 20.0, 1.0)], 20.0, 20.0, 0.0)
    add("no_obstacles", [], 20.0, 20.0, 1.0)
    add("ten_obstacles", [(22.0 + 1.7 * i, 20.0 + (-1) ** i * (1.0 + 0.3 * i), 0.9) for i in range(10)], 20.0, 20.0, 0.05)
    # wide theta sweep through a fixed obstacle field
    field = [(25.0, 25.0, 2.0), (18.0, 27.0, 1.5), (24.0, 16.0, 2.5), (14.0, 17.0, 1.2), (20.0, 29.5, 1.1)]
    for k in range(64):
        add(f"sweep_{k}", field, 20.0, 21.0, 2 * np.pi * k / 64.0)
    np.savez_compressed(
        os.path.join(OUT, "g4_sonar_edge.npz"),
        names=np.array([c["name"] for c in cases]),
        obs_tab=np.array([c["obs_tab"] for c in cases]),
        n_obs=np.array([c["n_obs"] for c in cases]),
        pose=np.array([c["pose"] for c in cases]),
        goal=np.array([c["goal"] for c in cases]),
        vel=np.array([c["vel"] for c in cases]),
        obs=np.array([c["obs"] for c in cases]),
    )


def g5_velocity():
    rng = np.random.RandomState(55)
    recs = {k: [] for k in ("cores", "n", "xy", "v")}
    for wi, nc in enumerate((8, 8, 4, 6, 1, 0, 2, 3)):
        env = MarineNavEnv(seed=500 + wi)
        env.num_cores, env.num_obs = nc, 0
        env.reset()
        c, _, n1, _ = world_arrays(env)
        pts = [rng.uniform(0, 50, size=2) for _ in range(40)]
        for k in env.cores:      # inside the core (d <= r) and just outside
            for rad in (0.05, 0.3, 0.4999, 0.5, 0.5001, 0.8):
                ang = rng.uniform(0, 2 * np.pi)
                pts.append(np.array([k.x + rad * np.cos(ang), k.y + rad * np.sin(ang)]))
        for p in pts:
            v = env.get_velocity(float(p[0]), float(p[1]))
            recs["cores"].append(c); recs["n"].append(n1); recs["xy"].append(p); recs["v"].append(v)
    np.savez_compressed(os.path.join(OUT, "g5_velocity.npz"), **{k: np.array(v) for k, v in recs.items()})


def g6_pretrained_replay():
    d = os.path.join(REF, "pretrained_models/IQN/seed_3")
    with open(os.path.join(d, "eval_config.json")) as f:
        cfg = json.load(f)
    out = {}
    for pol in ("greedy", "adaptive"):
        z = np.load(os.path.join(d, f"{pol}_evaluations.npz"), allow_pickle=True)
        evals = [0, 150, 299]
        envs = [0, 7, 13, 19, 24, 29]
        acts, lens, rews, succ, times, ener, ids = [], [], [], [], [], [], []
        for e in evals:
            for k in envs:
                a = np.asarray(z["actions"][e][k], dtype=np.int32)
                pad = np.full(1000, -1, dtype=np.int32)
                pad[: len(a)] = a
                acts.append(pad); lens.append(len(a))
                rews.append(float(z["rewards"][e][k])); succ.append(bool(z["successes"][e][k]))
                times.append(float(z["times"][e][k])); ener.append(float(z["energies"][e][k]))
                ids.append([e, k])
        out[f"{pol}_actions"] = np.array(acts); out[f"{pol}_len"] = np.array(lens)
        out[f"{pol}_reward"] = np.array(rews); out[f"{pol}_success"] = np.array(succ)
        out[f"{pol}_time"] = np.array(times); out[f"{pol}_energy"] = np.array(ener)
        out[f"{pol}_ids"] = np.array(ids)
        out[f"{pol}_timesteps_head"] = np.asarray(z["timesteps"][:3])
    np.savez_compressed(os.path.join(OUT, "g6_pretrained_replay.npz"), **out)
    # checkpoint data files (weights + constructor json) shipped by the reference
    pd = os.path.join(OUT, "pretrained_IQN_seed3")
    os.makedirs(pd, exist_ok=True)
    for fn in ("network_params.pth", "constructor_params.json"):
        shutil.copyfile(os.path.join(d, fn), os.path.join(pd, fn))
        os.chmod(os.path.join(pd, fn), 0o644)
    os.chmod(os.path.join(OUT, "eval_config_seed3.json"), 0o644)


def g7_iqn():
    import torch
    th = types.ModuleType("thirdparty"); th.__path__ = [os.path.join(REF, "thirdparty")]
    iq = types.ModuleType("thirdparty.IQN"); iq.__path__ = [os.path.join(REF, "thirdparty/IQN")]
    sys.modules["thirdparty"] = th; sys.modules["thirdparty.IQN"] = iq
    from thirdparty.IQN.agent import IQNAgent, calculate_huber_loss
    from thirdparty.IQN.model import ObsEncoder

    out = {}
    net = ObsEncoder(26, 9, seed=7)
    for k, v in net.state_dict().items():
        out["sd_" + k] = v.numpy().copy()
    # seeded-init parity: two nets with the same seed are identical
    rng = np.random.RandomState(9)
    obs = rng.normal(0, 5, size=(16, 26)).astype(np.float32)
    obs[:, 4:][rng.uniform(size=(16, 22)) < 0.5] = 0.0
    taus32 = rng.uniform(size=(16, 32)).astype(np.float32)
    taus8a = rng.uniform(size=(16, 8)).astype(np.float32)
    taus8b = rng.uniform(size=(16, 8)).astype(np.float32)

    inject = []
    real_rand = torch.rand

    def fake_rand(*shape, **kw):
        t = inject.pop(0)
        assert tuple(t.shape) == tuple(shape), (t.shape, shape)
        return torch.from_numpy(t.copy())

    torch.rand = fake_rand
    try:
        for cvar in (1.0, 0.5):
            inject.append(taus32)
            with torch.no_grad():
                q, t = net.forward(torch.from_numpy(obs), 32, cvar)
            out[f"fwd_quantiles_cvar{cvar}"] = q.numpy(); out[f"fwd_taus_cvar{cvar}"] = t.numpy()
            out[f"qvals_cvar{cvar}"] = q.mean(dim=1).numpy()
        # one train() step through the reference agent (B=16) with injected taus
        agent = IQNAgent(26, 9, BATCH_SIZE=16, seed=7)
        nxt = rng.normal(0, 5, size=(16, 26)).astype(np.float32)
        act = rng.randint(9, size=(16, 1)).astype(np.int64)
        rew = rng.normal(0, 3, size=(16, 1)).astype(np.float32)
        don = (rng.uniform(size=(16, 1)) < 0.25).astype(np.float32)
        # perturb the target net so the TD error is not trivially symmetric
        with torch.no_grad():
            for p in agent.qnetwork_target.parameters():
                p.add_(0.01 * torch.from_numpy(rng.normal(size=tuple(p.shape)).astype(np.float32)))
        for k, v in agent.qnetwork_target.state_dict().items():
            out["tgt_" + k] = v.numpy().copy()
        before = {k: v.numpy().copy() for k, v in agent.qnetwork_local.state_dict().items()}
        inject.extend([taus8a, taus8b])   # target forward draws first, then local (agent.py:279,285)
        exp = tuple(torch.from_numpy(x) for x in (obs, act, rew, nxt, don))
        loss = agent.train(exp)
        out["train_loss"] = np.array(loss)
        for (k, p) in agent.qnetwork_local.named_parameters():
            out["grad_" + k] = p.grad.numpy().copy()     # clipped in place by clip_grad_norm_
        for k, v in agent.qnetwork_local.state_dict().items():
            out["after_" + k] = v.numpy().copy()
            assert np.array_equal(before[k], out["sd_" + k])
    finally:
        torch.rand = real_rand
    out.update(obs=obs, taus32=taus32, taus8_target=taus8a, taus8_local=taus8b, next_obs=nxt,
               actions=act, rewards=rew, dones=don)
    # huber
    td = torch.from_numpy(rng.normal(0, 2, size=(4, 8, 8)).astype(np.float32))
    out["huber_in"] = td.numpy(); out["huber_out"] = calculate_huber_loss(td, 1.0).numpy()
    # adjust_cvar / linear_eps / energy cost tables
    ag = IQNAgent(26, 9, seed=0)
    states = rng.normal(0, 4, size=(32, 26))
    states[:8, 4:] = 0.0
    states[8:16, 4:] *= 0.0002
    out["cvar_states"] = states
    out["cvar_values"] = np.array([ag.adjust_cvar(s) for s in states])
    eps = []
    for t in (0, 1, 1000, 299_999, 300_000, 300_001, 3_000_000):
        ag.current_timestep = t
        eps.append(ag.linear_eps(3_000_000))
    out["eps_t"] = np.array([0, 1, 1000, 299_999, 300_000, 300_001, 3_000_000]); out["eps_v"] = np.array(eps)
    env = MarineNavEnv(seed=0)
    out["energy_cost"] = np.array([env.robot.compute_action_energy_cost(a) for a in range(9)])
    out["action_table"] = np.array(env.robot.actions)
    # pretrained net on the fixed obs with injected taus
    pre = ObsEncoder.load(os.path.join(REF, "pretrained_models/IQN/seed_3"))
    torch.rand = fake_rand
    try:
        inject.append(taus32)
        with torch.no_grad():
            q, _ = pre.forward(torch.from_numpy(obs), 32, 1.0)
        out["pretrained_quantiles"] = q.numpy()
    finally:
        torch.rand = real_rand
    np.savez_compressed(os.path.join(OUT, "g7_iqn.npz"), **out)


def _import_iqn():
    th = types.ModuleType("thirdparty"); th.__path__ = [os.path.join(REF, "thirdparty")]
    iq = types.ModuleType("thirdparty.IQN"); iq.__path__ = [os.path.join(REF, "thirdparty/IQN")]
    sys.modules["thirdparty"] = th; sys.modules["thirdparty.IQN"] = iq
    from thirdparty.IQN.agent import IQNAgent
    from thirdparty.IQN.replay_buffer import ReplayBuffer
    return IQNAgent, ReplayBuffer


def g12_replay():
    """The reference's ReplayBuffer (thirdparty/IQN/replay_buffer.py:6-59): 1500 adds into maxlen 1000, the deque
    contents afterwards (oldest -> newest), and what sample() returns (shapes, dtypes, which rows)."""
    _, ReplayBuffer = _import_iqn()
    rng = np.random.RandomState(12)
    n, cap, B = 1500, 1000, 32
    states = rng.normal(0, 5, size=(n, 26))
    nexts = rng.normal(0, 5, size=(n, 26))
    actions = rng.randint(9, size=n)
    rewards = rng.normal(0, 3, size=n)
    dones = rng.uniform(size=n) < 0.2
    buf = ReplayBuffer(cap, B, "cpu", seed=5, gamma=0.99)
    sizes = []
    for i in range(n):
        buf.add(states[i], int(actions[i]), float(rewards[i]), nexts[i], bool(dones[i]))
        sizes.append(len(buf))
    mem = list(buf.memory)
    out = dict(in_states=states, in_next=nexts, in_actions=actions, in_rewards=rewards, in_dones=dones,
               capacity=np.array(cap), batch=np.array(B), sizes=np.array(sizes),
               mem_states=np.stack([e.state for e in mem]), mem_next=np.stack([e.next_state for e in mem]),
               mem_actions=np.array([e.action for e in mem]), mem_rewards=np.array([e.reward for e in mem]),
               mem_dones=np.array([e.done for e in mem]))
    s, a, r, ns, d = buf.sample()
    out.update(sample_states=s.numpy(), sample_actions=a.numpy(), sample_rewards=r.numpy(), sample_next=ns.numpy(),
               sample_dones=d.numpy())
    for k in ("sample_states", "sample_actions", "sample_rewards", "sample_next", "sample_dones"):
        out[k + "_dtype"] = np.array(str(out[k].dtype))
    np.savez_compressed(os.path.join(OUT, "g12_replay.npz"), **out)


def g15_replay_nstep():
    """The reference's ReplayBuffer with n_step = 3 (replay_buffer.py:26-41): 40 adds into maxlen 25 -- the sliding window does not
    restart at `done` -- and the deque afterwards."""
    _, ReplayBuffer = _import_iqn()
    rng = np.random.RandomState(15)
    n_in, cap, n_step, gamma = 40, 25, 3, 0.97
    S = rng.normal(size=(n_in, 26)); NS = rng.normal(size=(n_in, 26))
    A = rng.randint(9, size=n_in); R = rng.normal(size=n_in); D = rng.uniform(size=n_in) < 0.2
    buf = ReplayBuffer(cap, 8, "cpu", seed=5, gamma=gamma, n_step=n_step)
    sizes = []
    for i in range(n_in):
        buf.add(S[i], int(A[i]), float(R[i]), NS[i], bool(D[i]))
        sizes.append(len(buf))
    mem = list(buf.memory)
    np.savez_compressed(os.path.join(OUT, "g15_replay_nstep.npz"), capacity=cap, n_step=n_step, gamma=gamma,
                        in_states=S, in_actions=A, in_rewards=R, in_next=NS, in_dones=D, sizes=np.array(sizes),
                        mem_states=np.stack([e.state for e in mem]), mem_actions=np.array([e.action for e in mem]),
                        mem_rewards=np.array([e.reward for e in mem]), mem_next=np.stack([e.next_state for e in mem]),
                        mem_dones=np.array([e.done for e in mem]))


def g13_learn_loop():
    """The reference's IQNAgent.learn (thirdparty/IQN/agent.py:94-173) run on the reference env for 400 timesteps:
    the trajectory-independent bookkeeping of the loop -- counters, at which learning steps train() / soft_update() /
    evaluation() fire, what the evaluation npz records -- for the drop-in loop to reproduce."""
    import tempfile
    import torch
    IQNAgent, _ = _import_iqn()
    cfg = dict(total_timesteps=400, learning_starts=100, eval_freq=150, target_update_interval=64, UPDATE_EVERY=4,
               BATCH_SIZE=32, BUFFER_SIZE=1000, seed=3, env_seed=11)
    torch.manual_seed(0)
    agent = IQNAgent(26, 9, BATCH_SIZE=cfg["BATCH_SIZE"], BUFFER_SIZE=cfg["BUFFER_SIZE"], UPDATE_EVERY=cfg["UPDATE_EVERY"],
                     learning_starts=cfg["learning_starts"], target_update_interval=cfg["target_update_interval"],
                     seed=cfg["seed"])
    train_env = MarineNavEnv(seed=cfg["env_seed"])
    eval_env = MarineNavEnv(seed=348)
    eval_env.reset_start_and_goal = False
    eval_env.start = np.array([5.0, 5.0]); eval_env.goal = np.array([45.0, 45.0])
    eval_config = {}
    for i, (nc, no) in enumerate(((4, 6), (8, 10))):
        eval_env.num_cores, eval_env.num_obs = nc, no
        quiet(eval_env.reset)
        eval_config[f"env_{i}"] = eval_env.episode_data()
    log = dict(train_at=[], train_mem=[], sync_at=[], eval_at=[], eval_ts=[], resets=0)
    real_train, real_sync, real_eval, real_reset = agent.train, agent.soft_update, agent.evaluation, train_env.reset

    def train(exp):
        log["train_at"].append(agent.learning_timestep); log["train_mem"].append(len(agent.memory))
        return real_train(exp)

    def sync(a, b):
        log["sync_at"].append(agent.learning_timestep)
        return real_sync(a, b)

    def evaluation(env, eval_config, greedy=True, eval_log_path=None):
        log["eval_at"].append((agent.learning_timestep, int(greedy))); log["eval_ts"].append(agent.current_timestep)
        return real_eval(env, eval_config=eval_config, greedy=greedy, eval_log_path=eval_log_path)

    def reset():
        log["resets"] += 1
        return real_reset()

    agent.train, agent.soft_update, agent.evaluation, train_env.reset = train, sync, evaluation, reset
    # third numpy-2 shim: agent.py:390 hands np.savez ragged python lists (per-episode action sequences), which numpy < 1.24
    # turned into object arrays silently and numpy 2 refuses
    real_savez = np.savez

    def savez(file, **kw):
        fixed = {}
        for k, v in kw.items():
            try:
                fixed[k] = np.asanyarray(v)
            except ValueError:
                fixed[k] = np.array(v, dtype=object)
        return real_savez(file, **fixed)

    np.savez = savez
    with tempfile.TemporaryDirectory() as tmp:
        quiet(agent.learn, total_timesteps=cfg["total_timesteps"], train_env=train_env, eval_env=eval_env,
              eval_config=eval_config, eval_freq=cfg["eval_freq"], eval_log_path=tmp, verbose=False)
        files = sorted(os.listdir(tmp))
        zg = np.load(os.path.join(tmp, "greedy_evaluations.npz"), allow_pickle=True)
        za = np.load(os.path.join(tmp, "adaptive_evaluations.npz"), allow_pickle=True)
        out = dict(npz_keys=np.array(sorted(zg.files)), greedy_timesteps=zg["timesteps"], adaptive_timesteps=za["timesteps"],
                   greedy_rewards_shape=np.array(zg["rewards"].shape), greedy_successes_shape=np.array(zg["successes"].shape))
    np.savez = real_savez
    out.update(files=np.array(files), current_timestep=np.array(agent.current_timestep),
               learning_timestep=np.array(agent.learning_timestep), train_at=np.array(log["train_at"]),
               train_mem=np.array(log["train_mem"]), sync_at=np.array(log["sync_at"]), eval_at=np.array(log["eval_at"]),
               eval_ts=np.array(log["eval_ts"]), memory_len=np.array(len(agent.memory)),
               env_total_timesteps=np.array(train_env.total_timesteps), train_resets=np.array(log["resets"]),
               eval_config=np.array(json.dumps(eval_config)), cfg=np.array(json.dumps(cfg)))
    np.savez_compressed(os.path.join(OUT, "g13_learn_loop.npz"), **out)


def g14_iqn_episodes():
    """run_experiments.py's evaluation_IQN loop (:19-72) restated around the reference's own env and agent
    (act_eval / act_adaptive_eval, pretrained seed_3 checkpoint) under exp_setup_5 (:192-211), with the taus of every
    act call injected from a seeded stream: per step the action, CVaR level, quantiles [1,32,9], taus [1,32,1]; per
    episode the outcome and episode_data()."""
    import torch
    IQNAgent, _ = _import_iqn()
    agent = IQNAgent(26, 9, seed=2)
    agent.load_model(os.path.join(REF, "pretrained_models/IQN/seed_3"), "cpu")
    real_rand = torch.rand
    out = {}
    for name, adaptive, cvar in (("adaptive", True, None), ("cvar0.5", False, 0.5), ("cvar1.0", False, 1.0)):
        env = MarineNavEnv(seed=15)
        env.reset_start_and_goal = False; env.random_reset_state = False; env.set_boundary = True
        env.obs_r_range = [1, 3]; env.start = np.array([5.0, 5.0]); env.goal = np.array([45.0, 45.0])
        env.robot.N = 5; env.num_cores, env.num_obs = 6, 8
        rng = np.random.RandomState(77)
        obs = env.reset()
        if name == "adaptive":
            c, o, n1, n2 = world_arrays(env)
            out.update(world_cores=c[:n1], world_obs=o[:n2], obs0=obs)
        rec = dict(actions=[], cvars=[], quantiles=[], taus=[], taus_in=[], obs=[])
        length, done, ret, energy = 0, False, 0.0, 0.0
        while not done and length < 1000:
            t_in = rng.uniform(size=(1, 32)).astype(np.float32)
            torch.rand = lambda *shape, **kw: torch.from_numpy(t_in.copy())
            try:
                if adaptive:
                    (action, quantiles, taus), cv = agent.act_adaptive_eval(obs)
                else:
                    action, quantiles, taus = agent.act_eval(obs, cvar=cvar)
                    cv = cvar
            finally:
                torch.rand = real_rand
            rec["obs"].append(obs); rec["taus_in"].append(t_in[0]); rec["actions"].append(int(action)); rec["cvars"].append(cv)
            rec["quantiles"].append(quantiles); rec["taus"].append(taus)
            obs, reward, done, info = env.step(int(action))
            ret += env.discount ** length * reward
            length += 1
            energy += env.robot.compute_action_energy_cost(int(action))
        ep = env.episode_data()
        for k, v in rec.items():
            out[f"{name}_{k}"] = np.array(v)
        out[f"{name}_success"] = np.array(info["state"] == "reach goal"); out[f"{name}_out_of_area"] = np.array(info["state"] == "out of boundary")
        out[f"{name}_time"] = np.array(env.robot.dt * env.robot.N * length); out[f"{name}_energy"] = np.array(energy)
        out[f"{name}_return"] = np.array(ret)
        out[f"{name}_trajectory"] = np.array(ep["robot"]["trajectory"])
        out[f"{name}_ep_keys"] = np.array(json.dumps({"env": sorted(ep["env"].keys()), "robot": sorted(ep["robot"].keys())}))
    np.savez_compressed(os.path.join(OUT, "g14_iqn_episodes.npz"), **out)


if __name__ == "__main__":
    which = sys.argv[1:] or ["g1", "g2", "g3", "g4", "g5", "g6", "g7", "g8", "g9", "g12", "g13", "g14", "g15"]
    if "g12" in which:
        g12_replay()
    if "g15" in which:
        g15_replay_nstep()
    if "g13" in which:
        g13_learn_loop()
    if "g14" in which:
        g14_iqn_episodes()
    if "g1" in which:
        g1_reset(); g1b_eval_worlds()
    if "g2" in which:
        g2_traces()
    if "g3" in which:
        g3_single_step()
    if "g4" in which:
        g4_sonar_edge()
    if "g5" in which:
        g5_velocity()
    if "g6" in which:
        g6_pretrained_replay()
    if "g7" in which:
        g7_iqn()
    if "g8" in which:
        g8_boundary_trace()
    if "g9" in which:
        g9_planners()
    for f in sorted(os.listdir(OUT)):
        p = os.path.join(OUT, f)
        if os.path.isfile(p):
            print(f"{os.path.getsize(p):>9d}  {f}")
