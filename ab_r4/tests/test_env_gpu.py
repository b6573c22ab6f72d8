"""GPU parity tests: the gfx950 kernels (through the C-ABI) against the CPU oracle and the golden
vectors generated from the Python reference.  Bit-exact for integer bookkeeping, world tables and
RNG stream position; float tolerances are written next to each check."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

G = os.path.join(os.path.dirname(__file__), "golden")
WORLD_SIZES = [(4, 6, 30.0), (6, 8, 35.0), (8, 10, 40.0), (8, 5, 25.0)]


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


def make_env(n, precision="f64", **kw):
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    # float64 copies of observations / rewards are opt-in (mn_enable_obs64): the parity tests read them
    kw.setdefault("obs64", precision == "f64")
    return VecMarineNavEnv(n, precision=precision, **kw)


def oracle_world(o):
    w = o.get_world()
    return w


def assert_world_equal(dev, orc):
    assert dev["n_cores"] == orc["n_cores"] and dev["n_obs"] == orc["n_obs"]
    assert np.array_equal(dev["cores"], orc["cores"])
    assert np.array_equal(dev["obstacles"], orc["obstacles"])
    assert np.array_equal(dev["start"], orc["start"]) and np.array_equal(dev["goal"], orc["goal"])
    assert dev["init_theta"] == orc["init_theta"] and dev["init_speed"] == orc["init_speed"]


@pytest.mark.parametrize("size", WORLD_SIZES)
def test_reset_bit_exact_vs_oracle(torch, size):
    """World generation + RNG stream position, 192 seeds x 3 consecutive resets."""
    from oracle.oracle import OracleEnv
    n = 192
    env = make_env(n, "f64", seed=0)
    env.set_attrs(num_cores=size[0], num_obs=size[1], min_start_goal_dis=size[2])
    orcs = [OracleEnv(i) for i in range(n)]
    for o in orcs:
        o.set_world_size(*size)
    for rep in range(3):
        env.reset()
        worlds = env.get_worlds()
        peek = env.peek_next_double()
        obs64 = env.get_obs64()
        st, ep, tot = env.get_state()
        for i, o in enumerate(orcs):
            oo = o.reset()
            assert_world_equal(worlds[i], o.get_world())
            assert peek[i] == o.peek_next_double(), (i, rep)
            np.testing.assert_allclose(obs64[i], oo, rtol=0, atol=1e-10)
            np.testing.assert_allclose(st[i], o.get_state()[0], rtol=0, atol=1e-12)
        assert (ep == 0).all()
    env.close()


def test_g1_golden_reset(torch):
    z = np.load(os.path.join(G, "g1_reset.npz"))
    i = 0
    while i < len(z["seed"]):
        env = make_env(1, "f64", seeds=[int(z["seed"][i])])
        nc, no, md = z["size"][i]
        env.set_attrs(num_cores=int(nc), num_obs=int(no), min_start_goal_dis=float(md))
        for k in range(3):
            j = i + k
            env.reset()
            w = env.get_worlds()[0]
            assert w["n_cores"] == z["ncores"][j] and w["n_obs"] == z["nobs"][j]
            assert np.array_equal(w["cores"], z["cores"][j][:w["n_cores"]])
            assert np.array_equal(w["obstacles"], z["obs"][j][:w["n_obs"]])
            assert np.array_equal(w["start"], z["start"][j]) and np.array_equal(w["goal"], z["goal"][j])
            assert w["init_theta"] == z["theta0"][j] and w["init_speed"] == z["speed0"][j]
            assert env.peek_next_double()[0] == z["next_double"][j]
            np.testing.assert_allclose(env.get_obs64()[0], z["obs0"][j], rtol=0, atol=1e-10)
        env.close()
        i += 3


def test_eval_config_regenerates_on_device(torch):
    """create_eval_configs (train_IQN_model.py:123-148): seed 348 reproduces the reference's shipped
    eval_config.json bit for bit."""
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    env = make_env(1, "f64", seeds=[348])
    env.set_attrs(reset_start_and_goal=False, obs_r_range=[1, 3])
    env.set_start_goal([5.0, 5.0], [45.0, 45.0])
    count = 0
    for nc, no in ((4, 6), (6, 8), (8, 10)):
        for _ in range(10):
            env.set_attrs(num_cores=nc, num_obs=no)
            env.reset()
            w = env.get_worlds()[0]
            e = cfg[f"env_{count}"]
            assert np.array_equal(w["cores"][:, :2], np.array(e["env"]["cores"]["positions"]))
            assert np.array_equal(w["cores"][:, 2], np.array(e["env"]["cores"]["clockwise"]))
            assert np.array_equal(w["cores"][:, 3], np.array(e["env"]["cores"]["Gamma"]))
            assert np.array_equal(w["obstacles"][:, :2], np.array(e["env"]["obstacles"]["positions"]))
            assert np.array_equal(w["obstacles"][:, 2], np.array(e["env"]["obstacles"]["r"]))
            assert w["init_theta"] == e["robot"]["init_theta"] and w["init_speed"] == e["robot"]["init_speed"]
            count += 1
    env.close()


@pytest.mark.parametrize("size", [(8, 10, 40.0), (4, 6, 30.0)])
def test_free_running_f64_vs_oracle(torch, size):
    """128 envs x 400 steps, random actions, auto-reset: float64 kernels follow the oracle through
    whole episodes; done/info/counters/worlds exact.  Floats <= 1e-6: libm-level differences (sincos
    ulp, FMA contraction) are amplified by the chaotic flow over hundreds of steps -- the reference
    itself drifts 1e-4 across numpy versions; the single-step bound (1e-9) is
    pinned by test_g3_single_step_golden."""
    from oracle.oracle import OracleEnv
    n, T = 128, 400
    env = make_env(n, "f64", seed=100)
    env.set_attrs(num_cores=size[0], num_obs=size[1], min_start_goal_dis=size[2])
    orcs = [OracleEnv(100 + i) for i in range(n)]
    for o in orcs:
        o.set_world_size(*size)
        o.reset()
    env.reset()
    rng = np.random.RandomState(5)
    worst = 0.0
    n_done = 0
    for t in range(T):
        a = rng.randint(9, size=n)
        env.step(torch.from_numpy(a).to(env.device))
        obs64 = env.get_obs64()
        rew = env.reward.cpu().numpy(); done = env.done.cpu().numpy(); info = env.info.cpu().numpy()
        rew64 = env.get_reward64()
        st, ep, tot = env.get_state()
        env.reset_done()
        robs = env.get_obs64()
        for i, o in enumerate(orcs):
            oo, r, d, inf = o.step(int(a[i]))
            assert d == bool(done[i]) and inf == info[i], (t, i)
            s, oep, otot = o.get_state()
            assert oep == ep[i] and otot == tot[i]
            worst = max(worst, np.abs(oo - obs64[i]).max(), abs(r - rew64[i]), np.abs(s - st[i]).max())
            assert abs(r - rew[i]) <= 1e-5     # float32 copy of the reward
            if d:
                n_done += 1
                ro = o.reset()
                worst = max(worst, np.abs(ro - robs[i]).max())
        assert worst < 1e-6, (t, worst)
    assert n_done > 5
    # worlds after all those resets are still bit-identical (RNG streams never drifted)
    worlds = env.get_worlds()
    peek = env.peek_next_double()
    for i, o in enumerate(orcs):
        assert_world_equal(worlds[i], o.get_world())
        assert peek[i] == o.peek_next_double()
    env.close()


def _miss(obs):
    """[n, 11] bool: beam reported as a miss, i.e. the point is exactly (0, 0) (marinenav_env.py:315-316)."""
    p = obs[:, 4:].reshape(len(obs), 11, 2)
    return (p[:, :, 0] == 0) & (p[:, :, 1] == 0)


def _load_g3(env, z, lo, hi):
    worlds = []
    for i in range(lo, hi):
        n1, n2 = z["n"][i]
        worlds.append(dict(cores=z["cores"][i][:n1], obstacles=z["obs_tab"][i][:n2], start=z["start"][i],
                           goal=z["goal"][i], init_theta=0.0, init_speed=0.0))
    env.load_worlds(worlds)
    s = np.zeros((hi - lo, 6))
    s[:, :4] = z["state_in"][lo:hi]
    env.set_state(s, z["ep_t"][lo:hi])


@pytest.mark.parametrize("lanes", [4, 1, 2, 8])
@pytest.mark.parametrize("precision,atol,rtol", [("f64", 1e-9, 0.0), ("mixed", 1e-5, 0.0)])
def test_g3_single_step_golden(torch, precision, atol, rtol, lanes):
    """2048 independent (world, state, action) triples from the Python reference.
    f64: <= 1e-9.  mixed: |err| <= 1e-5 ABSOLUTE on every float32 output (the north-star tolerance; f32 ulp at 50 m is
    3.8e-6) -- observation, pose, reward -- with NO outlier in this set, and identical discrete outcomes (done / info / beam hit
    or miss): the counts observed on MI355X, asserted as such."""
    z = np.load(os.path.join(G, "g3_single_step.npz"))
    n = len(z["action"])
    env = make_env(n, precision, step_lanes=lanes)   # lanes per env in the step kernel (default 4)
    _load_g3(env, z, 0, n)
    env.step(torch.from_numpy(z["action"].astype(np.int32)).to(env.device))
    obs = env.get_obs64() if precision == "f64" else env.obs.cpu().numpy().astype(np.float64)
    rew = env.reward.cpu().numpy().astype(np.float64)
    done = env.done.cpu().numpy().astype(bool); info = env.info.cpu().numpy()
    st = env.get_state()[0]
    if precision == "f64":
        assert np.array_equal(done, z["done"]) and np.array_equal(info, z["info"])
        np.testing.assert_allclose(obs[:, :4], z["obs"][:, :4], rtol=0, atol=atol)
        # The reference intersects beams in slope form (robot.py:164-179, K = tan(angle)), whose
        # rounding error grows like K^2; the kernel uses the well-conditioned ray form.  Allow the
        # reference's own conditioning: 1e-9 + 1e-12*K^2 (K reaches ~1e3 next to the snap window).
        theta = z["state_out"][:, 2]
        K = np.tan(theta[:, None] + (-np.pi / 3 + np.arange(11) * (2 * np.pi / 3) / 10)[None, :])
        tol = np.repeat(atol + 1e-12 * K * K, 2, axis=1)
        assert (np.abs(obs[:, 4:] - z["obs"][:, 4:]) <= tol).all()
        np.testing.assert_allclose(st, z["state_out"], rtol=0, atol=atol)
        np.testing.assert_allclose(rew, z["reward"], rtol=0, atol=1e-5)  # f32 output
    else:
        mism = np.nonzero(info != z["info"])[0]
        assert len(mism) == 0, mism            # observed on MI355X (r03): 0 info mismatches, 0 beam flips, 0 outliers, worst error 2.5e-6
        ok = np.ones(n, bool); ok[mism] = False
        # a beam may flip hit/miss when an intersection is within tolerance of the range / tangency
        beam_flip = _miss(obs) != _miss(z["obs"])
        assert beam_flip.sum() == 0
        keep = np.repeat(~beam_flip, 2, axis=1)
        err = np.abs(obs - z["obs"])
        err[:, 4:][~keep] = 0.0
        print(f"[observed g3 mixed lanes={lanes}] info mismatches {len(mism)}, beam flips {int(beam_flip.sum())}, outliers {int((err > atol).sum())}, worst {err.max():.3e}")
        assert (err > atol).sum() == 0, (int((err > atol).sum()), err.max())
        np.testing.assert_allclose(st, z["state_out"], rtol=0, atol=1.1e-5)       # velocity near a core edge: |v| ~ 10 m/s in float32
        np.testing.assert_allclose(st[:, :4], z["state_out"][:, :4], rtol=0, atol=atol)
        np.testing.assert_allclose(rew[ok], z["reward"][ok], rtol=0, atol=atol)
    env.close()


def test_loop_default_precision_is_strict_1e5_with_zero_outliers(torch):
    """The precision the training loop runs by default (bench.py with a learner, train_iqn, smoke): float64 env kernels,
    float32 outputs as the IQN consumes them.  North-star tolerance on the reference's 2048 single steps (G3): every
    float32 output (observation, reward) within 1e-5 ABSOLUTE, done / info identical -- no forgiveness clause."""
    import inspect
    import bench
    from distributional_rl_navigation_amd import train_iqn
    assert inspect.signature(train_iqn.run_trial).parameters["precision"].default == "f64"
    assert bench.default_precision(learner=True) == "f64" and bench.default_precision(learner=False) == "mixed"
    z = np.load(os.path.join(G, "g3_single_step.npz"))
    n = len(z["action"])
    env = make_env(n, bench.default_precision(learner=True))
    _load_g3(env, z, 0, n)
    env.step(torch.from_numpy(z["action"].astype(np.int32)).to(env.device))
    obs32 = env.obs.cpu().numpy()
    assert obs32.dtype == np.float32
    assert np.array_equal(env.done.cpu().numpy().astype(bool), z["done"]) and np.array_equal(env.info.cpu().numpy(), z["info"])
    assert np.array_equal(_miss(obs32.astype(np.float64)), _miss(z["obs"]))
    err = np.abs(obs32.astype(np.float64) - z["obs"])
    assert int((err > 1e-5).sum()) == 0, (int((err > 1e-5).sum()), err.max())
    assert np.abs(env.reward.cpu().numpy().astype(np.float64) - z["reward"]).max() <= 1e-5
    np.testing.assert_allclose(env.get_state()[0], z["state_out"], rtol=0, atol=1e-9)
    env.close()


def test_g4_sonar_edge_cases_on_device(torch):
    z = np.load(os.path.join(G, "g4_sonar_edge.npz"))
    n = len(z["names"])
    env = make_env(n, "f64")
    worlds = [dict(cores=np.zeros((0, 4)), obstacles=z["obs_tab"][i][:int(z["n_obs"][i])], start=z["pose"][i][:2],
                   goal=z["goal"][i], init_theta=float(z["pose"][i][2]), init_speed=1.0) for i in range(n)]
    env.load_worlds(worlds)
    obs = env.get_obs64()
    rel = -np.pi / 3 + np.arange(11) * (2 * np.pi / 3) / 10
    for i, name in enumerate(z["names"]):
        # obs[0:2] is the velocity (differs by construction: the golden case injects an arbitrary one)
        np.testing.assert_allclose(obs[i][2:4], z["obs"][i][2:4], rtol=0, atol=1e-9, err_msg=str(name))
        # slope-form conditioning of the reference (see test_g3): 1e-9 + 1e-12*tan(angle)^2
        K = np.tan(z["pose"][i][2] + rel)
        tol = np.repeat(1e-9 + 1e-12 * K * K, 2)
        assert (np.abs(obs[i][4:] - z["obs"][i][4:]) <= tol).all(), name
    env.close()


@pytest.mark.parametrize("policy", ["greedy", "adaptive"])
def test_g6_pretrained_replay_on_device(torch, policy):
    """Action sequences stored in the reference's own *_evaluations.npz replayed through its
    eval_config.json worlds: stored discounted return / success / time must come back."""
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    z = np.load(os.path.join(G, "g6_pretrained_replay.npz"))
    ids = z[f"{policy}_ids"]; L = z[f"{policy}_len"]; acts = z[f"{policy}_actions"]
    n = len(L)
    env = make_env(n, "f64")
    env.load_worlds([VecMarineNavEnv.world_from_eval_config(cfg[f"env_{k}"]) for _, k in ids])
    ret = np.zeros(n); alive = np.ones(n, bool); last_info = np.zeros(n, int); length = np.zeros(n, int)
    for t in range(int(L.max())):
        a = np.where(acts[:, t] >= 0, acts[:, t], 0).astype(np.int32)
        env.step(torch.from_numpy(a).to(env.device))
        r = env.get_reward64(); d = env.done.cpu().numpy().astype(bool)
        inf = env.info.cpu().numpy()
        live = alive & (t < L)
        ret[live] += 0.99 ** t * r[live]
        length[live] += 1
        last_info[live] = inf[live]
        alive &= ~(d & live)
    assert np.array_equal(length, L)
    assert np.array_equal(last_info == 4, z[f"{policy}_success"])
    np.testing.assert_allclose(0.1 * 10 * length, z[f"{policy}_time"], atol=1e-9)
    # float64 rewards: the reference's stored returns come back to 1e-9 (1e-6 on the 1000-step
    # episodes, where libm-level differences are amplified by the chaotic flow -- SURVEY section 4)
    tol = np.where(L > 600, 1e-6, 1e-9)
    assert (np.abs(ret - z[f"{policy}_reward"]) <= tol).all(), np.abs(ret - z[f"{policy}_reward"]).max()
    env.close()


def test_mixed_single_step_vs_oracle_states(torch):
    """Mixed precision, every step restarted from the float64 trajectory of the f64 kernels, 200 steps x 1024 envs
    (5.3 M float32 outputs): ABSOLUTE error <= 1e-5 on every output except a handful of conditioning outliers, < 1e-4
    (measured over 10.6 M outputs, scripts/diag_mixed.py: 6 above 1e-5, worst 4e-5 -- grazing sonar returns, where the
    ~1e-7 m pose error of the float32 current field is multiplied by r/h, and the velocity itself within 0.3 m of a
    vortex core edge); pose x / y / heading / speed <= 1e-5 always; zero beam or done flips expected, <= 20 allowed."""
    n, T = 1024, 200
    e64 = make_env(n, "f64", seed=7)
    emx = make_env(n, "mixed", seed=7)
    for e in (e64, emx):
        e.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        e.reset()
    rng = np.random.RandomState(3)
    flips, outliers, worst = 0, 0, 0.0
    for t in range(T):
        a = torch.from_numpy(rng.randint(9, size=n).astype(np.int32)).to(e64.device)
        s, ep, tot = e64.get_state()
        emx.set_state(s, ep, tot)
        e64.step(a); emx.step(a)
        o64 = e64.get_obs64(); omx = emx.obs.cpu().numpy().astype(np.float64)
        d64 = e64.done.cpu().numpy(); dmx = emx.done.cpu().numpy()
        bad = d64 != dmx
        flips += int(bad.sum())
        beam_flip = _miss(o64) != _miss(omx)
        flips += int(beam_flip.sum())
        keep = np.concatenate([np.ones((n, 4), bool), np.repeat(~beam_flip, 2, axis=1)], axis=1)
        err = np.abs(o64 - omx)
        outliers += int((err[keep] > 1e-5).sum()); worst = max(worst, float(err[keep].max()))
        smx = emx.get_state()[0]; s64 = e64.get_state()[0]
        np.testing.assert_allclose(smx[:, :4], s64[:, :4], rtol=0, atol=1e-5)
        outliers += int((np.abs(smx[:, 4:] - s64[:, 4:]) > 1e-5).sum()); worst = max(worst, float(np.abs(smx[:, 4:] - s64[:, 4:]).max()))
        r64 = e64.reward.cpu().numpy(); rmx = emx.reward.cpu().numpy()
        np.testing.assert_allclose(rmx[~bad], r64[~bad], rtol=0, atol=1e-5)
        e64.reset_done()
        # worlds must stay identical: give the mixed env the same resets
        emx.reset(mask=e64.done)
    print(f"[observed mixed vs f64] flips {flips}, outliers {outliers}, worst {worst:.3e}")
    # observed on MI355X (r03): 0 flips, 1 outlier of 1.3e-5 in 5.3 M outputs; the strict float64 kernels (the loop default) have none
    assert flips == 0, flips
    assert outliers <= 3 and worst < 5e-5, (outliers, worst)
    w64 = e64.get_worlds(); wmx = emx.get_worlds()
    for a_, b_ in zip(w64, wmx):
        assert_world_equal(a_, b_)
    e64.close(); emx.close()


@pytest.mark.parametrize("precision", ["f64", "mixed"])
def test_full_size_properties(torch, precision):
    """65 536 envs (BASELINE config size): shard equivalence + invariants (size-independent), in the precision the training loop /
    bench.py runs by default (float64, `bench.default_precision`) and in the kernel-only configs' mixed precision."""
    n, sub, T = 65536, 2048, 60
    big = make_env(n, precision, seed=0, obs64=False)
    small = make_env(sub, precision, seed=0, first_index=0, obs64=False)
    small2 = make_env(sub, precision, seed=0, first_index=n - sub, obs64=False)   # last shard of the big run
    for e in (big, small, small2):
        e.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        e.reset()
    g = torch.Generator(device=big.device); g.manual_seed(0)
    total_done = 0
    for t in range(T):
        a = torch.randint(0, 9, (n,), device=big.device, dtype=torch.int32, generator=g)
        big.step_autoreset(a); small.step_autoreset(a[:sub]); small2.step_autoreset(a[n - sub:])
        assert torch.equal(big.obs[:sub], small.obs) and torch.equal(big.obs[n - sub:], small2.obs)
        assert torch.equal(big.reward[:sub], small.reward) and torch.equal(big.done[n - sub:], small2.done)
        d = big.done.bool()
        assert torch.isfinite(big.obs).all() and torch.isfinite(big.reward).all()
        assert ((big.info != 0) == d).all()
        total_done += int(d.sum())
        assert big.last_done_count() == int(d.sum())
    assert total_done > 0
    s, ep, tot = big.get_state()
    assert (tot == T).all() and (ep <= T).all() and (ep >= 0).all()
    w = big.get_worlds(0, 512)
    assert all(x["n_cores"] <= 8 and x["n_obs"] <= 10 for x in w)
    for e in (big, small, small2):
        e.close()


def test_full_size_loop_kernel_against_the_oracle(torch):
    """The instantiation behind bench.py's `value` -- `mn_step_kernel<double, true, 4, APPEND>` at 65 536 envs, i.e. `mn_step_append` on a
    float64 handle + `mn_reset_done` -- for 20 vector steps with the replay append, rows [0, 256) and [65 280, 65 536) followed by 512
    scalar oracle envs (marinenav_env.py:199-262 restated in oracle/marinenav_oracle.c): done / info / counters exact, float32
    observations and rewards within 1e-5 absolute (the north-star bound), the ring rows of those envs hold the same transitions."""
    from oracle.oracle import OracleEnv
    from distributional_rl_navigation_amd.iqn.replay_buffer import ReplayBuffer
    n, T, cap = 65536, 20, 100_000
    rows = np.r_[0:256, n - 256:n]
    env = make_env(n, "f64", seed=0, obs64=False)      # exactly the loop's handle: no float64 copies
    env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    buf = ReplayBuffer(cap, 256, env.device, seed=0, gamma=0.99)
    orcs = [OracleEnv(int(i)) for i in rows]
    oobs = []
    for o in orcs:
        o.set_world_size(8, 10, 40.0)
        oobs.append(o.reset())
    obs = env.reset()
    np.testing.assert_allclose(obs[rows].cpu().numpy(), np.array(oobs), rtol=0, atol=1e-5)
    g = torch.Generator(device=env.device); g.manual_seed(3)
    worst, n_done = 0.0, 0
    for t in range(T):
        a = torch.randint(0, 9, (n,), device=env.device, dtype=torch.int32, generator=g)
        prev = obs.clone()
        ptr0 = buf.ptr
        nobs, rew, done, info = env.step_append(a, obs, buf)
        ah = a[rows].cpu().numpy(); oh = nobs[rows].cpu().numpy(); rh = rew[rows].cpu().numpy()
        dh = done[rows].cpu().numpy(); ih = info[rows].cpu().numpy()
        st, ep, tot = env.get_state()
        # ring rows of the compared envs (slot = (ptr + e) mod cap; n < cap)
        slots = torch.from_numpy((ptr0 + rows) % cap).to(env.device)
        assert torch.equal(buf.states[slots], prev[rows]) and torch.equal(buf.next_states[slots], nobs[rows])
        assert torch.equal(buf.rewards[slots, 0], rew[rows]) and torch.equal(buf.dones[slots, 0], done[rows].float())
        assert torch.equal(buf.actions[slots, 0].to(torch.int32), a[rows])
        obs = env.reset_done()
        robs = obs[rows].cpu().numpy()
        for k, o in enumerate(orcs):
            oo, r, d, inf = o.step(int(ah[k]))
            assert d == bool(dh[k]) and inf == ih[k], (t, k)
            s, oep, otot = o.get_state()
            assert oep == ep[rows[k]] and otot == tot[rows[k]]
            worst = max(worst, float(np.abs(oo - oh[k]).max()), abs(r - float(rh[k])))
            if d:
                n_done += 1
                worst = max(worst, float(np.abs(o.reset() - robs[k]).max()))
        assert worst <= 1e-5, (t, worst)
    assert n_done > 0
    w = env.get_worlds(0, 256) + env.get_worlds(n - 256, 256)
    for k, o in enumerate(orcs):
        assert_world_equal(w[k], o.get_world())
    env.close()


def test_obs64_copies_are_opt_in_and_change_nothing_else(torch):
    """`mn_enable_obs64`: the float64 observation / reward copies are written only when switched on; every other output of the
    float64 kernels -- observations, rewards, done / info, poses, counters, worlds, replay ring -- is bit-identical either way."""
    from distributional_rl_navigation_amd.iqn.replay_buffer import ReplayBuffer
    n, cap = 3000, 8192
    envs = [make_env(n, "f64", seed=9, obs64=flag) for flag in (False, True)]
    bufs = [ReplayBuffer(cap, 32, envs[0].device, seed=0, gamma=0.99) for _ in range(2)]
    obs = []
    for e in envs:
        e.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        obs.append(e.reset())
    with pytest.raises(Exception):
        envs[0].get_obs64()
    g = torch.Generator(device=envs[0].device); g.manual_seed(2)
    for t in range(30):
        a = torch.randint(0, 9, (n,), device=envs[0].device, dtype=torch.int32, generator=g)
        outs = [e.step_append(a, o, b) for e, o, b in zip(envs, obs, bufs)]
        for x, y in zip(*outs):
            assert torch.equal(x, y)
        np.testing.assert_allclose(envs[1].get_obs64(), outs[1][0].cpu().numpy().astype(np.float64), rtol=0, atol=1e-5)
        obs = [e.reset_done() for e in envs]
        assert torch.equal(obs[0], obs[1])
    for x, y in ((bufs[0].states, bufs[1].states), (bufs[0].next_states, bufs[1].next_states), (bufs[0].rewards, bufs[1].rewards)):
        assert torch.equal(x, y)
    assert all(np.array_equal(a_, b_) for a_, b_ in zip(envs[0].get_state(), envs[1].get_state()))
    # switched on later: valid from the next step on; switched off again: the getter refuses
    envs[0].enable_obs64(True)
    a = torch.zeros(n, dtype=torch.int32, device=envs[0].device)
    envs[0].step(a); envs[1].step(a)
    assert np.array_equal(envs[0].get_obs64(), envs[1].get_obs64()) and np.array_equal(envs[0].get_reward64(), envs[1].get_reward64())
    envs[0].enable_obs64(False)
    with pytest.raises(Exception):
        envs[0].get_reward64()
    mixed = make_env(64, "mixed", seed=0)
    with pytest.raises(Exception):
        mixed.enable_obs64(True)
    for e in envs + [mixed]:
        e.close()


def test_schedule_and_ragged_batch(torch):
    """Curriculum lookup (marinenav_env.py:89-98) against the golden schedule trace, on a batch
    whose size is not a multiple of the 64-env tile."""
    z = np.load(os.path.join(G, "g2_trace_seed5_schedule.npz"))
    sched = dict(timesteps=z["sched_timesteps"], num_cores=z["sched_num_cores"],
                 num_obstacles=z["sched_num_obstacles"], min_start_goal_dis=z["sched_min_dis"])
    n = 67
    seeds = np.full(n, int(z["seed"]), dtype=np.uint32)
    env = make_env(n, "f64", seeds=seeds, schedule=sched)
    env.reset()
    np.testing.assert_allclose(env.get_obs64()[[0, 63, 64, 66]], np.tile(z["obs0"], (4, 1)), atol=1e-10)
    wi = 0
    worst = 0.0
    for t, a in enumerate(z["actions"]):
        env.step(torch.full((n,), int(a), dtype=torch.int32, device=env.device))
        o = env.get_obs64()
        assert np.abs(o - o[0]).max() == 0.0           # identical seeds -> identical lanes
        worst = max(worst, np.abs(o[66] - z["obs"][t]).max())
        assert bool(env.done[66].item()) == bool(z["done"][t]) and int(env.info[66].item()) == int(z["info"][t]), t
        if z["done"][t]:
            env.reset_done()
            wi += 1
            w = env.get_worlds(66, 1)[0]
            assert [w["n_cores"], w["n_obs"]] == list(z["world_n"][wi])
            assert np.array_equal(w["cores"], z["world_cores"][wi][:w["n_cores"]])
            assert np.array_equal(w["obstacles"], z["world_obs"][wi][:w["n_obs"]])
            np.testing.assert_allclose(env.get_obs64()[66], z["reset_obs"][t], atol=1e-10)
    s, ep, tot = env.get_state()
    assert ep[66] == z["ep_t"][-1] and tot[66] == z["tot_t"][-1]
    assert worst < 1e-7, worst
    env.close()


def test_g8_boundary_and_robot_n5_on_device(torch):
    """Out-of-boundary branch (marinenav_env.py:240-243) and robot.N = 5 (run_experiments.py:204) through
    the gym-shaped facade, against the reference trace."""
    from distributional_rl_navigation_amd.marinenav_env.env import MarineNavEnv
    z = np.load(os.path.join(G, "g8_boundary_trace.npz"))
    env = MarineNavEnv(seed=int(z["seed"]))
    env.set_boundary = True
    env.robot.N = 5
    env.reset_start_and_goal = False
    env.start = np.array(z["start"]); env.goal = np.array(z["goal"])
    env.num_cores, env.num_obs = 8, 8
    np.testing.assert_allclose(env.reset(), z["obs0"], atol=1e-10)
    names = ("normal", "out of boundary", "too long episode", "collision", "reach goal")
    for t, a in enumerate(z["actions"]):
        obs, r, d, info = env.step(int(a))
        assert d == bool(z["done"][t]) and info["state"] == names[int(z["info"][t])], t
        np.testing.assert_allclose(obs, z["obs"][t], atol=1e-7)
        assert abs(r - z["reward"][t]) < 1e-7
        if d:
            np.testing.assert_allclose(env.reset(), z["reset_obs"][t], atol=1e-10)
    assert (z["info"] == 1).sum() >= 5
    env.close()


def test_c_abi_error_behaviour(torch):
    """Status codes instead of exceptions across the C-ABI: over-capacity worlds, bad parameters,
    out-of-range env windows; mn_last_error carries the text."""
    import ctypes as C
    from distributional_rl_navigation_amd import _capi
    L = _capi.lib()
    p = _capi.default_params()
    p.num_beams = 12
    h = C.c_void_p()
    assert L.mn_create(4, C.byref(p), C.byref(h)) == -1 and b"num_beams" in L.mn_last_error(None)
    assert L.mn_create(0, C.byref(_capi.default_params()), C.byref(h)) == -1
    env = make_env(4, "mixed")
    with pytest.raises(ValueError):
        env.load_worlds([dict(cores=np.zeros((9, 4)), obstacles=np.zeros((0, 3)), start=[1, 1], goal=[2, 2],
                              init_theta=0.0, init_speed=0.0)])
    with pytest.raises(_capi.MarineNavHipError):
        env.set_attrs(num_cores=9)                       # beyond the device capacity of 8
    env.params.num_cores = 8
    with pytest.raises(_capi.MarineNavHipError):
        env.get_state(2, 5)                              # window past n_envs
    with pytest.raises(_capi.MarineNavHipError):
        env.get_obs64()                                  # float64 copies exist only in f64 precision
    with pytest.raises(_capi.MarineNavHipError):
        env.set_schedule(dict(timesteps=[0], num_cores=[9], num_obstacles=[1], min_start_goal_dis=[30.0]))
    # actions outside [0, 9) are clamped, never read out of bounds
    env.set_attrs(num_cores=4, num_obs=6)
    env.reset()
    env.step(torch.tensor([-5, 100, 3, 8], dtype=torch.int32, device=env.device))
    assert bool(torch.isfinite(env.obs).all())
    env.close()


def test_vector_curriculum_timestep_scale(torch):
    """Curriculum in vector mode: the stage is looked up with total_timesteps[i] * timestep_scale
    (marinenav_env.py:89-98 with aggregate experience), so with scale = n_envs the stages of
    train_IQN_model.py:86-90 switch after the same number of ENV steps as in the reference."""
    n = 8
    sched = dict(timesteps=[0, 80, 160], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10], min_start_goal_dis=[30.0, 35.0, 40.0])
    env = make_env(n, "mixed", seed=3, schedule=sched, timestep_scale=n)
    env.reset()
    assert all(w["n_cores"] <= 4 and w["n_obs"] <= 6 for w in env.get_worlds())
    a = torch.zeros(n, dtype=torch.int32, device=env.device)
    for t in range(1, 25):
        env.step(a)
        env.reset(mask=torch.ones(n, dtype=torch.uint8, device=env.device))      # force a reset every step
        w = env.get_worlds()
        agg = t * n                                                              # aggregate env steps so far
        stage = 0 if agg < 80 else (1 if agg < 160 else 2)
        want_c, want_o = sched["num_cores"][stage], sched["num_obstacles"][stage]
        assert all(x["n_cores"] <= want_c and x["n_obs"] <= want_o for x in w), (t, stage)
        assert any(x["n_cores"] == want_c for x in w) and any(x["n_obs"] >= want_o - 2 for x in w), (t, stage)
        prev = sched["num_cores"][stage - 1] if stage else 0
        assert max(x["n_cores"] for x in w) > prev
    env.close()


def test_c_abi_from_plain_c(torch, tmp_path):
    """The boundary is a C ABI: examples/c_abi_demo.c (gcc, no Python, no torch) creates, resets and steps
    4096 envs through libmarinenav_hip.so and checks the counters."""
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    pkg = os.path.join(root, "distributional_rl_navigation_amd")
    exe = str(tmp_path / "c_abi_demo")
    subprocess.check_call(["gcc", "-D__HIP_PLATFORM_AMD__", os.path.join(root, "examples", "c_abi_demo.c"),
                           "-I" + os.path.join(root, "include"), "-I/opt/rocm/include", "-L" + pkg, "-lmarinenav_hip",
                           "-L/opt/rocm/lib", "-lamdhip64", "-Wl,-rpath," + pkg, "-Wl,-rpath,/opt/rocm/lib", "-o", exe])
    out = subprocess.check_output([exe, "4096", "120"], text=True)
    assert "total_timesteps 120" in out and "M env steps/s" in out, out


@pytest.mark.parametrize("precision", ["f64", "mixed"])
def test_results_do_not_depend_on_lanes_per_env(torch, precision):
    """`step_lanes` is a performance knob only: the eight vortex contributions are added in one fixed balanced tree and
    every fused multiply-add is written out (csrc/mn_device.h), so 1, 2, 4 and 8 lanes per env give BIT-identical
    observations, rewards, poses -- which is what lets a 65 536-env shard (2 lanes) equal its slice of a 524 288-env run
    (1 lane)."""
    n, T = 3000, 40
    envs = [make_env(n, precision, seed=11, step_lanes=L) for L in (1, 2, 4, 8)]
    obs = []
    for e in envs:
        e.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        obs.append(e.reset().clone())
    for o in obs[1:]:
        assert torch.equal(o, obs[0])
    g = torch.Generator(device=envs[0].device); g.manual_seed(2)
    for t in range(T):
        a = torch.randint(0, 9, (n,), device=envs[0].device, dtype=torch.int32, generator=g)
        outs = [tuple(x.clone() for x in e.step(a)) for e in envs]
        for o in outs[1:]:
            assert all(torch.equal(x, y) for x, y in zip(o, outs[0])), t
        for e in envs:
            e.reset_done()
    ref = envs[0].get_state()
    for e in envs[1:]:
        assert all(np.array_equal(x, y) for x, y in zip(e.get_state(), ref))
    if precision == "f64":
        for e in envs[1:]:
            assert np.array_equal(e.get_obs64(), envs[0].get_obs64())
    for e in envs:
        e.close()
