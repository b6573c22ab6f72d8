"""`mn_rollout`: T vector steps (in-kernel action draws, in-kernel reset of finished envs) in one launch must be
BIT-IDENTICAL to T x (mn_random_actions -> mn_step -> mn_reset_done) -- observations, rewards, done / info codes,
actions, final pose, counters, world tables and RNG stream positions -- in float64 and in mixed precision, for every
lanes-per-env mapping of either kernel; and shard == slice for the in-kernel action draws."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


def _make(n, precision, **kw):
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(n, seed=5, device="cuda:0", precision=precision, obs64=precision == "f64", **kw)
    env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    return env


def _single_launches(torch, env, T, seed, step0=0, actions=None):
    """The reference sequence: one launch per step, reset of finished envs between steps."""
    obs, rew, done, info, act = [], [], [], [], []
    for t in range(T):
        a = env.random_actions(seed, step0 + t) if actions is None else actions[t]
        o, r, d, i = env.step(a)
        obs.append(o.clone()); rew.append(r.clone()); done.append(d.clone()); info.append(i.clone()); act.append(a.clone())
        env.reset_done()
    return dict(obs=torch.stack(obs), reward=torch.stack(rew), done=torch.stack(done), info=torch.stack(info),
                action=torch.stack(act), final_obs=env.obs.clone())


def _same_state(a, b):
    sa, sb = a.get_state(), b.get_state()
    assert all(np.array_equal(x, y) for x, y in zip(sa, sb))
    assert np.array_equal(a.peek_next_double(), b.peek_next_double())          # RNG stream positions
    for wa, wb in zip(a.get_worlds(0, 64), b.get_worlds(0, 64)):
        assert np.array_equal(wa["cores"], wb["cores"]) and np.array_equal(wa["obstacles"], wb["obstacles"])
        assert np.array_equal(wa["start"], wb["start"]) and np.array_equal(wa["goal"], wb["goal"])


@pytest.mark.parametrize("precision", ["f64", "mixed"])
@pytest.mark.parametrize("n,T,rl,sl", [(700, 120, 0, 0), (700, 120, 2, 8), (700, 120, 4, 1), (700, 120, 8, 2), (700, 120, 16, 2), (64, 300, 16, 4), (64, 300, 8, 4), (5000, 40, 0, 0), (20000, 12, 0, 0)])
def test_rollout_equals_single_launches(torch, precision, n, T, rl, sl):
    a_env = _make(n, precision, rollout_lanes=rl, step_lanes=sl)      # one launch
    b_env = _make(n, precision, rollout_lanes=rl, step_lanes=sl)      # T x (actions, step, reset_done)
    assert torch.equal(a_env.reset(), b_env.reset())
    ref = _single_launches(torch, b_env, T, seed=42)
    out = a_env.rollout(T, action_seed=42, trace=("obs", "reward", "done", "info", "action"))
    for k in ("obs", "reward", "done", "info", "action"):
        assert torch.equal(out[k], ref[k]), k
    assert torch.equal(out["final_obs"], ref["final_obs"])
    assert int(ref["done"].sum()) >= min(n // 8, 200)                                   # plenty of in-kernel resets were compared
    _same_state(a_env, b_env)
    if precision == "f64":
        assert np.array_equal(a_env.get_obs64(), b_env.get_obs64()) and np.array_equal(a_env.get_reward64(), b_env.get_reward64())
    # the two entry points keep interleaving: a second rollout continues the step counter, then single steps again
    out2 = a_env.rollout(7, action_seed=42, first_step=T, trace=("obs", "done"))
    ref2 = _single_launches(torch, b_env, 7, seed=42, step0=T)
    assert torch.equal(out2["obs"], ref2["obs"]) and torch.equal(out2["done"], ref2["done"])
    assert a_env.last_done_count() == 0                                        # nothing pending after a rollout
    a = a_env.random_actions(42, T + 7)
    oa = a_env.step(a)[0].clone(); ob = b_env.step(a)[0].clone()
    assert torch.equal(oa, ob)
    assert torch.equal(a_env.reset_done(), b_env.reset_done())
    _same_state(a_env, b_env)
    a_env.close(); b_env.close()


def test_rollout_with_given_actions_and_curriculum(torch):
    """Caller-supplied [T, n] actions, the curriculum schedule (timestep_scale) consulted by the in-kernel resets, and
    the minimal-trace call (no traces at all)."""
    n, T = 512, 200
    sched = dict(timesteps=[0, 60, 120], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10], min_start_goal_dis=[30.0, 35.0, 40.0])
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    envs = [VecMarineNavEnv(n, seed=9, device="cuda:0", precision="f64", schedule=sched, timestep_scale=1.0) for _ in range(2)]
    assert torch.equal(envs[0].reset(), envs[1].reset())
    g = torch.Generator(device="cuda:0"); g.manual_seed(3)
    acts = torch.randint(0, 9, (T, n), device="cuda:0", dtype=torch.int32, generator=g)
    ref = _single_launches(torch, envs[1], T, seed=0, actions=acts)
    out = envs[0].rollout(T, actions=acts, trace=())
    assert set(out.keys()) == {"final_obs"} and torch.equal(out["final_obs"], ref["final_obs"])
    _same_state(envs[0], envs[1])
    w = envs[0].get_worlds()
    assert max(x["n_cores"] for x in w) == 8 and min(x["n_cores"] for x in w) >= 4      # later stages were reached in-kernel
    for e in envs:
        e.close()


def test_rollout_shards_equal_slices(torch):
    """Action draws are keyed by the GLOBAL env index: a shard (first_index = r * n) rolls out exactly its slice of
    the big run (BASELINE configs[1] x world size)."""
    n, world, T = 1024, 4, 50
    big = _make(n * world, "mixed")
    big.reset()
    ob = big.rollout(T, action_seed=7, trace=("obs", "reward", "done", "action"))
    for r in range(world):
        sh = _make(n, "mixed", first_index=r * n)
        sh.reset()
        o = sh.rollout(T, action_seed=7, trace=("obs", "reward", "done", "action"))
        sl = slice(r * n, (r + 1) * n)
        for k in ("obs", "reward", "done", "action"):
            assert torch.equal(o[k], ob[k][:, sl]), (r, k)
        assert torch.equal(o["final_obs"], ob["final_obs"][sl])
        sh.close()
    a = ob["action"]
    hist = torch.bincount(a.reshape(-1).long(), minlength=9).float() / a.numel()
    assert float((hist - 1 / 9).abs().max()) < 0.01                                   # uniform over the 9 actions
    big.close()


def test_rollout_argument_checks(torch):
    import ctypes as C
    from distributional_rl_navigation_amd import _capi
    env = _make(64, "mixed")
    o = env.reset()
    L = _capi.lib()
    p = lambda t: C.c_void_p(t.data_ptr())
    assert L.mn_rollout(env.h, 0, None, 0, 0, 0, p(o), None, None, None, None, None, None) == -1        # n_steps < 1
    assert L.mn_rollout(env.h, 4, None, 0, 0, 0, None, None, None, None, None, None, None) == -1       # obs_dev required
    assert L.mn_rollout(None, 4, None, 0, 0, 0, p(o), None, None, None, None, None, None) == -1
    assert L.mn_random_actions(0, 0, 0, 0, p(o), None) == -1 and L.mn_random_actions(0, 0, 0, 4, None, None) == -1
    env.params.rollout_lanes = 1      # no 1-lane rollout variant
    assert L.mn_set_params(env.h, C.byref(env.params)) == -1 and b"rollout_lanes" in L.mn_last_error(env.h)
    env.close()
