"""Pins oracle/marinenav_oracle.c against golden vectors generated from the Python reference
(tests/golden/make_golden.py) and the reference's own shipped evaluation artefacts."""
import json
import os

import numpy as np
import pytest

from oracle.oracle import OracleEnv

G = os.path.join(os.path.dirname(__file__), "golden")


def test_g1_reset_bit_exact():
    z = np.load(os.path.join(G, "g1_reset.npz"))
    n = len(z["seed"])
    i = 0
    while i < n:
        seed = int(z["seed"][i])
        nc, no, md = z["size"][i]
        env = OracleEnv(seed)
        env.set_world_size(nc, no, md)
        for k in range(3):
            obs = env.reset()
            w = env.get_world()
            j = i + k
            assert w["n_cores"] == z["ncores"][j] and w["n_obs"] == z["nobs"][j]
            assert np.array_equal(w["start"], z["start"][j]) and np.array_equal(w["goal"], z["goal"][j])
            assert np.array_equal(w["cores"], z["cores"][j][: w["n_cores"]])
            assert np.array_equal(w["obstacles"], z["obs"][j][: w["n_obs"]])
            assert w["init_theta"] == z["theta0"][j] and w["init_speed"] == z["speed0"][j]
            assert env.peek_next_double() == z["next_double"][j]      # RNG stream position
            np.testing.assert_allclose(obs, z["obs0"][j], rtol=0, atol=1e-11)
            np.testing.assert_allclose(env.get_state()[0], z["state0"][j], rtol=0, atol=1e-12)
        i += 3


def test_eval_config_regenerates_bit_exact():
    """train_IQN_model.py:123-148 with seed 348 must reproduce the shipped eval_config.json."""
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    env = OracleEnv(348)
    env.set_flags(reset_start_and_goal=False)
    env.set_start_goal([5.0, 5.0], [45.0, 45.0])
    count = 0
    for nc, no in ((4, 6), (6, 8), (8, 10)):
        for _ in range(10):
            env.set_world_size(nc, no, 25.0)
            env.reset()
            w = env.get_world()
            e = cfg[f"env_{count}"]
            assert np.array_equal(w["cores"][:, :2], np.array(e["env"]["cores"]["positions"]))
            assert np.array_equal(w["cores"][:, 2], np.array(e["env"]["cores"]["clockwise"]))
            assert np.array_equal(w["cores"][:, 3], np.array(e["env"]["cores"]["Gamma"]))
            assert np.array_equal(w["obstacles"][:, :2], np.array(e["env"]["obstacles"]["positions"]))
            assert np.array_equal(w["obstacles"][:, 2], np.array(e["env"]["obstacles"]["r"]))
            assert w["init_theta"] == e["robot"]["init_theta"] and w["init_speed"] == e["robot"]["init_speed"]
            count += 1
    assert count == 30


def _replay_trace(fn):
    z = np.load(os.path.join(G, fn))
    sched = None
    if "sched_timesteps" in z.files:
        sched = dict(timesteps=z["sched_timesteps"], num_cores=z["sched_num_cores"],
                     num_obstacles=z["sched_num_obstacles"], min_start_goal_dis=z["sched_min_dis"])
    env = OracleEnv(int(z["seed"]), sched)
    if sched is None:
        env.set_world_size(*z["size"])
    obs = env.reset()
    np.testing.assert_allclose(obs, z["obs0"], rtol=0, atol=1e-10)
    wi = 0
    worst = 0.0
    for t, a in enumerate(z["actions"]):
        obs, r, d, info = env.step(int(a))
        s, ep_t, tot_t = env.get_state()
        assert d == bool(z["done"][t]) and info == z["info"][t], (fn, t)
        assert ep_t == z["ep_t"][t] and tot_t == z["tot_t"][t]
        worst = max(worst, np.abs(obs - z["obs"][t]).max(), abs(r - z["reward"][t]), np.abs(s - z["state"][t]).max())
        if d:
            ro = env.reset()
            wi += 1
            w = env.get_world()
            assert [w["n_cores"], w["n_obs"]] == list(z["world_n"][wi])
            assert np.array_equal(w["cores"], z["world_cores"][wi][: w["n_cores"]])
            assert np.array_equal(w["obstacles"], z["world_obs"][wi][: w["n_obs"]])
            assert np.array_equal(w["start"], z["world_start"][wi]) and np.array_equal(w["goal"], z["world_goal"][wi])
            np.testing.assert_allclose(ro, z["reset_obs"][t], rtol=0, atol=1e-10)
    return worst


@pytest.mark.parametrize("fn", ["g2_trace_seed0_default.npz", "g2_trace_seed1_stage0.npz",
                                "g2_trace_seed2_stage2.npz", "g2_trace_seed5_schedule.npz"])
def test_g2_free_running_traces(fn):
    worst = _replay_trace(fn)
    assert worst < 1e-8, worst


def test_g3_single_step():
    z = np.load(os.path.join(G, "g3_single_step.npz"))
    env = OracleEnv(0)
    worst = 0.0
    for i in range(len(z["action"])):
        n1, n2 = z["n"][i]
        env.load_world(z["cores"][i], n1, z["obs_tab"][i], n2, z["start"][i], z["goal"][i], 0.0, 0.0)
        s = np.zeros(6)
        s[:4] = z["state_in"][i]
        env.set_state(s, int(z["ep_t"][i]))
        obs, r, d, info = env.step(int(z["action"][i]))
        assert d == bool(z["done"][i]) and info == z["info"][i], i
        so = env.get_state()[0]
        worst = max(worst, np.abs(obs - z["obs"][i]).max(), abs(r - z["reward"][i]), np.abs(so - z["state_out"][i]).max())
    assert worst < 1e-9, worst


def test_g4_sonar_edge_cases():
    z = np.load(os.path.join(G, "g4_sonar_edge.npz"))
    env = OracleEnv(0)
    for i, name in enumerate(z["names"]):
        env.load_world(np.zeros((0, 4)), 0, z["obs_tab"][i], int(z["n_obs"][i]), [0, 0], z["goal"][i], 0.0, 0.0)
        x, y, th = z["pose"][i]
        env.set_state([x, y, th, 1.0, z["vel"][i][0], z["vel"][i][1]])
        obs = env.get_observation()
        np.testing.assert_allclose(obs, z["obs"][i], rtol=0, atol=1e-9, err_msg=str(name))


def test_g5_velocity():
    z = np.load(os.path.join(G, "g5_velocity.npz"))
    env = OracleEnv(0)
    for i in range(len(z["n"])):
        env.load_world(z["cores"][i], int(z["n"][i]), np.zeros((0, 3)), 0, [0, 0], [1, 1], 0.0, 0.0)
        v = env.get_velocity(*z["xy"][i])
        np.testing.assert_allclose(v, z["v"][i], rtol=0, atol=1e-12)


@pytest.mark.parametrize("policy,tol", [("greedy", 1e-9), ("adaptive", 1e-9)])
def test_g6_pretrained_replay(policy, tol):
    """Replays action sequences stored in the reference's own *_evaluations.npz through the
    eval worlds of its eval_config.json; expected values are the reference's stored ones."""
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    z = np.load(os.path.join(G, "g6_pretrained_replay.npz"))
    env = OracleEnv(0)
    for i in range(len(z[f"{policy}_len"])):
        _, k = z[f"{policy}_ids"][i]
        env.load_eval_config(cfg[f"env_{k}"])
        L = int(z[f"{policy}_len"][i])
        ret, info, done = 0.0, 0, False
        for t in range(L):
            assert not done
            _, r, done, info = env.step(int(z[f"{policy}_actions"][i][t]))
            ret += 0.99 ** t * r
        assert done or L == 1000
        assert (info == 4) == bool(z[f"{policy}_success"][i])
        assert abs(0.1 * 10 * L - z[f"{policy}_time"][i]) < 1e-9
        assert abs(ret - z[f"{policy}_reward"][i]) < max(tol, 1e-6 if L > 600 else tol), (i, ret, z[f"{policy}_reward"][i])


def test_g8_boundary_and_robot_n5_trace():
    """set_boundary = True + robot.N = 5 (run_experiments.py:192-211 settings): out-of-boundary branch of
    the termination ladder (marinenav_env.py:240-243) against the reference."""
    z = np.load(os.path.join(G, "g8_boundary_trace.npz"))
    env = OracleEnv(int(z["seed"]))
    env.set_flags(reset_start_and_goal=False, random_reset_state=True, set_boundary=True)
    env.set_robot_N(5)
    env.set_start_goal(z["start"], z["goal"])
    env.set_world_size(8, 8, 25.0)
    np.testing.assert_allclose(env.reset(), z["obs0"], atol=1e-10)
    worst = 0.0
    for t, a in enumerate(z["actions"]):
        obs, r, d, info = env.step(int(a))
        assert d == bool(z["done"][t]) and info == z["info"][t], t
        assert env.get_state()[1] == z["ep_t"][t]
        worst = max(worst, np.abs(obs - z["obs"][t]).max(), abs(r - z["reward"][t]))
        if d:
            np.testing.assert_allclose(env.reset(), z["reset_obs"][t], atol=1e-10)
    assert (z["info"] == 1).sum() >= 5 and worst < 1e-8
