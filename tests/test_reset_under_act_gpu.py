"""Episode resets UNDER the next vector step's act kernel (C-ABI mn_reset_done_async / mn_reset_join / mn_iqn_set_late_rows;
`VecMarineNavEnv.reset_done(under_next_act=True)`, `IQNAgent.reset_under_act`): the reset launch runs on the env handle's own stream while
the act kernel -- told which rows are being rewritten -- takes those rows last, each after its "row is final" word.  Same results as the
reset IN FRONT of the act kernel (agent.py:113-124: `env.reset()` then `act(state)`), bit for bit, whatever share of the rows is late."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


def _loop(torch, n, precision, under_act, T, max_episode_steps=None, shared_taus=False, eps=0.3, train=True):
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(n, seed=3, device=DEV, precision=precision)
    if max_episode_steps is not None:
        env.params.max_episode_steps = max_episode_steps
    env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    agent = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=3 * n, device=DEV, seed=11, learning_starts=0, UPDATE_EVERY=2 if train else 10 ** 9)
    agent.shared_taus = shared_taus
    agent.reset_under_act = under_act
    env.set_reset_under_act_max(2 ** 31 - 1)      # however many episodes end (the library's default goes in front above a decaying peak of 1 200 per vector step)
    obs = env.reset()
    trace = []
    for t in range(T):
        prev_late = env.late_rows is not None
        obs, reward, done, info, loss = agent.vec_step(env, obs, eps)
        assert (env.late_rows is not None) == (under_act and not shared_taus)      # the reset of THIS step is pending / is not
        if t and under_act and not shared_taus:
            assert prev_late                                   # ... and the previous one was consumed by this step's act launch
        trace.append((reward.clone(), done.clone(), info.clone(), None if loss is None else float(loss)))
    env.join_reset()
    agent.check_learner()
    m = agent.memory
    out = dict(trace=trace, obs=obs.clone(), ring=(m.states.clone(), m.actions.clone(), m.rewards.clone(), m.next_states.clone(), m.dones.clone()),
               params=torch.cat([p.detach().reshape(-1).clone() for p in agent.qnetwork_local.parameters()]), state=env.get_state(),
               dones=int(sum(int(tr[1].sum()) for tr in trace)))
    from distributional_rl_navigation_amd.iqn.fused_act import late_timeouts
    assert late_timeouts(agent.qnetwork_local) == 0
    env.close()
    return out


def _same(torch, a, b):
    for (r0, d0, i0, l0), (r1, d1, i1, l1) in zip(a["trace"], b["trace"]):
        assert torch.equal(r0, r1) and torch.equal(d0, d1) and torch.equal(i0, i1) and l0 == l1
    assert torch.equal(a["obs"], b["obs"]) and torch.equal(a["params"], b["params"])
    for x, y in zip(a["ring"], b["ring"]):
        assert torch.equal(x, y)
    for x, y in zip(a["state"], b["state"]):
        assert np.array_equal(x, y)


@pytest.mark.parametrize("n,precision", [(4096, "f64"), (5000, "mixed"), (65536, "f64")])
def test_loop_with_resets_under_the_act_kernel_equals_the_plain_loop(torch, n, precision):
    """act -> step + append -> reset -> (train) for 40 vector steps with the resets in front of the act kernel and under it: rewards, done /
    info codes, losses, observations, replay ring, learned parameters, env state and counters identical."""
    a = _loop(torch, n, precision, False, 40)
    b = _loop(torch, n, precision, True, 40)
    assert a["dones"] > 0
    _same(torch, a, b)


def test_every_row_late(torch):
    """max_episode_steps = 3: ALL envs finish in the same vector step, so every row of the following act launch is a late row (and every
    wavefront of the reset launch has to find room beside the act kernel's workgroups)."""
    a = _loop(torch, 8192, "f64", False, 10, max_episode_steps=3, train=False)
    b = _loop(torch, 8192, "f64", True, 10, max_episode_steps=3, train=False)
    assert a["dones"] >= 2 * 8192
    _same(torch, a, b)


@pytest.mark.parametrize("precision", ["f64", "mixed"])
def test_reset_with_the_rng_block_in_device_memory_equals_the_lds_form(torch, precision):
    """The under-act reset kernel reads each env's MT19937 row in place and regenerates it out of registers (MtInPlace, csrc/mn_reset_body.h) where the
    kernel in front of the act launch copies it into LDS.  12 consecutive resets of every env (max_episode_steps = 1: each one consumes ~300 of the 624
    words of a block, so every env crosses ~5 block boundaries, with 0-6 words carried over) through both: worlds, poses, first observations and the
    stream position (the next double every env would draw) identical."""
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    outs = []
    for under in (False, True):
        env = VecMarineNavEnv(3000, seed=17, device=DEV, precision=precision)
        env.params.max_episode_steps = 1
        env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
        env.set_reset_under_act_max(2 ** 31 - 1)
        obs = env.reset()
        a = torch.zeros(3000, dtype=torch.int32, device=DEV)
        first = []
        for t in range(12):
            env.step(a)
            env.step(a)                      # the second step of an episode times out: every env is done
            assert int(env.done.sum()) == 3000
            obs = env.reset_done(under_next_act=under)
            assert (env.late_rows is not None) == under
            env.join_reset()
            torch.cuda.synchronize()
            first.append(obs.clone())
        worlds = env.get_worlds()
        outs.append((torch.stack(first), env.peek_next_double(), env.get_state(), worlds))
        assert env.reset_launches == ([0, 12] if under else [0, 0])      # (counts reset_done(under_next_act=True) calls: [ran in front, ran under])
        env.close()
    (o0, p0, s0, w0), (o1, p1, s1, w1) = outs
    assert torch.equal(o0, o1) and np.array_equal(p0, p1)
    for x, y in zip(s0, s1):
        assert np.array_equal(x, y)
    for x, y in zip(w0, w1):
        assert np.array_equal(x["cores"], y["cores"]) and np.array_equal(x["obstacles"], y["obstacles"])


def test_forms_without_late_rows_keep_the_reset_in_front(torch):
    """Launch-shared taus (another kernel form) cannot take late rows: vec_step leaves the reset in front (`late_rows_possible`), and an act call that
    is handed a pending reset it cannot honour joins it first -- same results as the plain loop either way."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    a = _loop(torch, 4096, "f64", False, 20, shared_taus=True)
    b = _loop(torch, 4096, "f64", True, 20, shared_taus=True)
    _same(torch, a, b)
    # the join path: a pending reset handed to an act launch of the shared-tau form / of the exact-f32 variant
    outs = []
    for pending in (False, True):
        env = VecMarineNavEnv(2048, seed=5, device=DEV, precision="f64")
        env.set_reset_under_act_max(2 ** 31 - 1)
        agent = IQNAgent(26, 9, device=DEV, seed=4)
        agent.shared_taus = True
        obs = env.reset()
        acts = []
        for t in range(12):
            a_ = agent.act_batch(obs, 0.2, late_env=env)
            assert env.late_rows is None
            acts.append(a_.clone())
            env.step(a_)
            obs = env.reset_done(under_next_act=pending)
        env.join_reset()
        outs.append((torch.stack(acts), obs.clone()))
        env.close()
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])


def _learn_vec_run(torch, under, steps, max_episode_steps=None, on_step=None, before=None):
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(4096, seed=2, device=DEV, precision="f64")
    if max_episode_steps is not None:
        env.params.max_episode_steps = max_episode_steps
    env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    agent = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=3 * 4096, device=DEV, seed=5, learning_starts=0, UPDATE_EVERY=2)
    if before is not None:
        before(env, agent)
    agent.learn_vec(total_vector_steps=steps, train_env=env, verbose=False, reset_under_act=under,
                    on_step=(lambda it, st: on_step(it, env, agent)) if on_step else None)
    out = dict(params=torch.cat([p.detach().reshape(-1).clone() for p in agent.qnetwork_local.parameters()]), obs=env.obs.clone(), state=env.get_state(),
               launches=list(env.reset_launches), fallback=agent.under_act_fallback, max_after=env.reset_under_act_max, flag_after=agent.reset_under_act)
    env.close()
    return out


def test_learn_vec_preflight_healthy_box(torch):
    """learn_vec's first 12 vector steps force every reset under the act kernel and read the late-row time-outs (UnderActGuard): on a box where the reset
    launch runs beside the act kernel nothing happens -- no fallback, the library's rule restored, same parameters / observations / env state as the loop
    with the resets in front."""
    a = _learn_vec_run(torch, False, 40, max_episode_steps=5)
    b = _learn_vec_run(torch, True, 40, max_episode_steps=5)
    assert b["fallback"] is None and a["fallback"] is None
    assert b["launches"][1] == 40 and a["launches"] == [0, 0]
    assert b["max_after"] == 5000 and b["flag_after"] is False      # (the agent's own default is restored behind the loop)
    assert torch.equal(a["params"], b["params"]) and torch.equal(a["obs"], b["obs"])
    for x, y in zip(a["state"], b["state"]):
        assert np.array_equal(x, y)


def test_learn_vec_falls_back_when_the_reset_does_not_run_beside_the_act_kernel(torch):
    """The no-co-residency case, simulated: a kernel that sleeps 20 ms in front of every reset launch on the env's own stream (mn_debug_side_delay_us), late rows
    that wait 1 ms at most (mn_iqn_set_late_bound_ms).  Every env finishes at every step (max_episode_steps = 0), so the second act launch of the loop finds all
    its rows unfinished: the preflight sees the time-outs after vector step 1, says so once, and the loop carries on with the resets in front of the act kernel --
    no exception at the end, nothing under the act kernel after the fallback."""
    from distributional_rl_navigation_amd.iqn.fused_act import set_late_bound_ms, late_timeouts

    def hold(env, agent):
        set_late_bound_ms(agent.qnetwork_local, 1.0)
        env.debug_side_delay_us(20000)
    r = _learn_vec_run(torch, True, 30, max_episode_steps=0, before=hold)
    assert r["fallback"] is not None and r["fallback"]["step"] == 1 and r["fallback"]["timeouts"] > 0
    assert r["launches"] == [0, 2]                      # vector steps 0 and 1 went under the act kernel, none after
    assert r["max_after"] == 5000
    assert torch.isfinite(r["params"]).all()

    # ... and when it starts to happen in the middle of a run (after the preflight): found by the unsynchronised look every 64 vector steps
    def later(it, env, agent):
        if it == 20:
            set_late_bound_ms(agent.qnetwork_local, 1.0)
            env.debug_side_delay_us(20000)
    r = _learn_vec_run(torch, True, 150, max_episode_steps=0, on_step=later)
    # (the look does not synchronise: it sees what the launches EXECUTED so far have reported -- normally at the first poll behind step 20, vector step 75; a host that ran
    # far ahead of the device would find it at the next one, 139)
    assert r["fallback"] is not None and r["fallback"]["step"] in (75, 139) and r["fallback"]["timeouts"] > 0
    assert r["launches"] == [0, r["fallback"]["step"] + 1]


def test_check_learner_raises_for_callers_without_the_guard(torch):
    """A caller that drives vec_step itself with reset_under_act and never looks gets the old behaviour: check_learner() raises."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.iqn.fused_act import set_late_bound_ms
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(4096, seed=2, device=DEV, precision="f64")
    env.params.max_episode_steps = 0
    env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    agent = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=3 * 4096, device=DEV, seed=5, learning_starts=0, UPDATE_EVERY=10 ** 9)
    agent.reset_under_act = True
    set_late_bound_ms(agent.qnetwork_local, 1.0)
    env.debug_side_delay_us(20000)
    obs = env.reset()
    for _ in range(3):
        obs = agent.vec_step(env, obs, 0.5)[0]
    env.join_reset()
    with pytest.raises(RuntimeError, match="before their episode reset had finished"):
        agent.check_learner()
    env.close()


def test_c_abi_contract(torch):
    import ctypes as C
    from distributional_rl_navigation_amd import _capi
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.iqn.fused_act import act_context
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    L = _capi.lib()
    env = VecMarineNavEnv(256, seed=0, device=DEV, precision="f64")
    env.reset()
    env.set_reset_under_act_max(2 ** 31 - 1)
    ready, tick = C.c_void_p(), C.c_uint32()
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert L.mn_reset_done_async(None, p(env.obs), s, C.byref(ready), C.byref(tick)) != 0
    assert L.mn_reset_done_async(env.h, None, s, C.byref(ready), C.byref(tick)) != 0
    assert L.mn_reset_done_async(env.h, p(env.obs), s, None, C.byref(tick)) != 0
    assert L.mn_reset_join(None, s) != 0 and L.mn_reset_join(env.h, s) == 0      # nothing pending: a no-op
    env.step(torch.zeros(256, dtype=torch.int32, device=DEV))
    assert L.mn_reset_done_async(env.h, p(env.obs), s, C.byref(ready), C.byref(tick)) == 0 and ready.value and tick.value == 1
    s0, e0, t0 = env.get_state()      # a host accessor with a reset pending: waits for it
    assert L.mn_reset_join(env.h, s) == 0
    agent = IQNAgent(26, 9, device=DEV, seed=1)
    ctx = act_context(agent.qnetwork_local)
    assert L.mn_iqn_set_late_rows(None, p(env.done), ready, 1, 256) != 0
    assert L.mn_iqn_set_late_rows(ctx.h, p(env.done), None, 1, 256) < 0
    assert L.mn_iqn_set_late_rows(ctx.h, p(env.done), ready, 1, 256) == 0
    assert L.mn_iqn_set_late_rows(ctx.h, None, None, 0, 0) == 0               # clear
    assert L.mn_iqn_set_late_rows(ctx.h, p(env.done), ready, 1, 64 * 8 * 256 + 8) == 1      # more than 64 rows per wavefront: join instead
    ctx.set_variant(0)
    assert L.mn_iqn_set_late_rows(ctx.h, p(env.done), ready, 1, 256) == 1     # the exact-f32 kernel has no such form
    ctx.set_variant(2)
    out = C.c_uint32(7)
    assert L.mn_iqn_late_timeouts(ctx.h, s, C.byref(out)) == 0 and out.value == 0
    out = C.c_uint32(7)
    assert L.mn_iqn_late_timeouts_peek(ctx.h, C.byref(out)) == 0 and out.value == 0
    assert L.mn_iqn_late_timeouts_peek(None, C.byref(out)) != 0 and L.mn_iqn_late_timeouts_peek(ctx.h, None) != 0
    assert L.mn_iqn_set_late_bound_ms(ctx.h, C.c_double(0.0)) != 0 and L.mn_iqn_set_late_bound_ms(ctx.h, C.c_double(1e9)) != 0
    assert L.mn_iqn_set_late_bound_ms(None, C.c_double(1.0)) != 0 and L.mn_iqn_set_late_bound_ms(ctx.h, C.c_double(500.0)) == 0
    assert L.mn_debug_side_delay_us(None, 1) != 0 and L.mn_debug_side_delay_us(env.h, -1) != 0 and L.mn_debug_side_delay_us(env.h, 0) == 0
    env.close()


def test_many_episode_ends_go_in_front(torch):
    """The library's own rule: while the decaying peak (x 7/8 per call) of the episode counts of the reset launches seen is above `under_act_max`, the next
    mn_reset_done_async runs in front of the act kernel (ready_out NULL, no late rows) -- and goes back under it when the counts have fallen."""
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    n = 4096
    env = VecMarineNavEnv(n, seed=0, device=DEV, precision="f64")
    env.params.max_episode_steps = 2
    env.set_attrs(num_cores=4, num_obs=6, min_start_goal_dis=30.0)
    assert env.set_reset_under_act_max(1000) == -1      # nothing seen yet
    env.reset()
    a = torch.zeros(n, dtype=torch.int32, device=DEV)
    modes, counts = [], []
    for t in range(6):
        if t == 3:
            env.params.max_episode_steps = 1000      # from here on (almost) no episode ends
            env.set_attrs(num_cores=4, num_obs=6)
        env.step(a)
        env.reset_done(under_next_act=True)
        modes.append(env.late_rows is not None)
        env.join_reset()
        torch.cuda.synchronize()
        counts.append(env.set_reset_under_act_max(1000))      # the launches' decaying peak of their episode counts
        if t <= 2:
            assert counts[-1] == int(env.done.sum())
    for t in range(30):
        env.step(a); env.reset_done(under_next_act=True); modes.append(env.late_rows is not None); env.join_reset(); torch.cuda.synchronize()
    # step 0: nothing seen -> in front (0 episodes seen afterwards); steps 1, 2: under; all 4 096 episodes end in step 2 (max_episode_steps = 2: the third step
    # of an episode); from then on the peak decays from 4 096 and is <= 1 000 after 11 more launches
    assert modes[:3] == [False, True, True] and counts[:4] == [0, 0, n, n - n // 8] and not any(modes[3:13]) and all(modes[16:]), (modes, counts)
    env.close()
