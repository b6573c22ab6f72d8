"""Fused IQN action-value kernel (csrc/iqn_act.hip) against the plain PyTorch float32 ObsEncoder."""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


@pytest.mark.parametrize("n", [1, 7, 64, 1000, 8192 + 3])
@pytest.mark.parametrize("weights", ["seeded", "pretrained"])
def test_fused_qvals_matches_torch(torch, n, weights):
    """Same taus, same weights: |fused - torch| <= 2e-5 + 1e-6*max|Q| (|Q| reaches ~600 on these synthetic
    observations: a few float32 ulps of the largest activations).  Measured
    (scripts/act_accuracy.py, pretrained net, 16 384 real observations): both paths sit 2.1e-5 from a float64
    evaluation and 2.3e-5 from each other -- float32 chains of ~600 terms in different summation orders."""
    from distributional_rl_navigation_amd.iqn.fused_act import fused_qvals
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    if weights == "seeded":
        net = ObsEncoder(26, 9, seed=11, device="cuda:0")
    else:
        net = ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
    g = torch.Generator(device="cuda:0"); g.manual_seed(n)
    obs = torch.randn(n, 26, device="cuda:0", generator=g) * 6.0
    obs[:, 4:] = torch.where(torch.rand(n, 22, device="cuda:0", generator=g) < 0.5, torch.zeros(()).cuda(), obs[:, 4:])
    taus = torch.rand(n, 32, device="cuda:0", generator=g)
    for cvar in (1.0, 0.37):
        with torch.no_grad():
            ref = net.get_qvals(obs, cvar, taus=taus)
        out = fused_qvals(net, obs, cvar, taus=taus)
        err = (out - ref).abs()
        tol = 2e-5 + 1e-6 * ref.abs().max()
        assert bool((err <= tol).all()), (float(err.max()), float(ref.abs().max()))
    # per-row cvar tensor
    cv = torch.rand(n, device="cuda:0", generator=g)
    with torch.no_grad():
        ref = net.get_qvals(obs, cv, taus=taus)
    out = fused_qvals(net, obs, cv, taus=taus)
    assert bool(((out - ref).abs() <= 2e-5 + 1e-6 * ref.abs().max()).all())


def test_fused_act_agrees_with_torch_argmax(torch):
    """Greedy actions from the fused path equal the PyTorch path wherever the top-2 Q gap exceeds the
    numerical tolerance (pretrained policy, real observations from the env)."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    agent = IQNAgent(26, 9, device="cuda:0", seed=0, BUFFER_SIZE=1024)
    agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
    env = VecMarineNavEnv(4096, seed=3, device="cuda:0")
    obs = env.reset()
    taus = torch.rand(4096, 32, device="cuda:0")
    with torch.no_grad():
        q_ref = agent.qnetwork_local.get_qvals(obs, 1.0, taus=taus)
    q = agent.qvals_batch(obs, 1.0, taus=taus)
    top2 = q_ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-3
    assert bool((q.argmax(1)[clear] == q_ref.argmax(1)[clear]).all())
    assert float(clear.float().mean()) > 0.9
    env.close()


def test_replay_append_kernel_matches_torch_ring(torch):
    """mn_replay_append (one launch) == ReplayBuffer.add_batch (indexed copies), incl. wrap-around and
    n > capacity (deque(maxlen) semantics, replay_buffer.py:19,26-34)."""
    from distributional_rl_navigation_amd.iqn.replay_buffer import ReplayBuffer
    g = torch.Generator(device="cuda:0"); g.manual_seed(5)
    a = ReplayBuffer(1000, 32, "cuda:0", seed=0, gamma=0.99)
    b = ReplayBuffer(1000, 32, "cuda:0", seed=0, gamma=0.99)
    for n in (300, 300, 300, 300, 64, 1500, 7, 1000):
        obs = torch.randn(n, 26, device="cuda:0", generator=g)
        nxt = torch.randn(n, 26, device="cuda:0", generator=g)
        act = torch.randint(0, 9, (n,), device="cuda:0", dtype=torch.int32, generator=g)
        rew = torch.randn(n, device="cuda:0", generator=g)
        done = (torch.rand(n, device="cuda:0", generator=g) < 0.1).to(torch.uint8)
        a.add_vector_step(obs, act, rew, nxt, done)
        b.add_batch(obs, act.long(), rew, nxt, done.float())
        assert a.ptr == b.ptr and a.size == b.size
        for x, y in ((a.states, b.states), (a.next_states, b.next_states), (a.actions, b.actions),
                     (a.rewards, b.rewards), (a.dones, b.dones)):
            assert torch.equal(x, y), n


def test_graphed_train_step_equals_eager(torch):
    """(opt-in path, IQNAgent.use_train_graph) The hipGraph-replayed grad step (forward, backward, clip, Adam) == the eager one: same batches,
    same generator state -> same taus -> same weights after 5 steps."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    g = torch.Generator(device="cuda:0"); g.manual_seed(2)
    batches = []
    for _ in range(5):
        batches.append((torch.randn(64, 26, device="cuda:0", generator=g) * 4, torch.randint(0, 9, (64, 1), device="cuda:0", generator=g),
                        torch.randn(64, 1, device="cuda:0", generator=g), torch.randn(64, 26, device="cuda:0", generator=g) * 4,
                        (torch.rand(64, 1, device="cuda:0", generator=g) < 0.2).float()))
    res = []
    for graphed in (False, True):
        ag = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=256, device="cuda:0", seed=9)
        ag.use_fused_train = False       # this test is about the two PyTorch paths
        ag.use_train_graph = graphed
        torch.manual_seed(1234)
        if graphed:                      # building the graph draws taus during warm-up: build first,
            ag.train(batches[0])         # then restart from the same weights / generator state
            ag2 = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=256, device="cuda:0", seed=9)
            ag.qnetwork_local.load_state_dict(ag2.qnetwork_local.state_dict())
            ag.qnetwork_target.load_state_dict(ag2.qnetwork_target.state_dict())
            with torch.no_grad():            # the captured graph updates THIS optimizer's state tensors: back to a fresh Adam, in place
                for st in ag.optimizer.state.values():
                    for v_ in st.values():
                        if torch.is_tensor(v_):
                            v_.zero_()
            torch.manual_seed(1234)
        losses = [float(ag.train(b)) for b in batches]
        res.append((losses, [p.detach().clone() for p in ag.qnetwork_local.parameters()]))
    (l0, p0), (l1, p1) = res
    # Same taus: the first losses agree to float rounding (different taus would differ by percents).
    # Later steps drift by O(lr): Adam turns rounding-level differences of near-zero gradients into
    # +-lr updates, so weights are compared at a few lr (1e-4) and losses at 2e-3.
    np.testing.assert_allclose(l0[:2], l1[:2], rtol=1e-5)
    np.testing.assert_allclose(l0, l1, rtol=2e-3)
    for a, b in zip(p0, p1):
        assert float((a - b).abs().max()) < 6e-4


def test_fused_act_epilogue_argmax_and_exploration(torch):
    """Epilogue of the act kernel (agent.py:199-203): greedy = first argmax of its own Q-values;
    eps-greedy takes the greedy action iff u > eps, else a uniform random action."""
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    net = ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
    n = 20000
    g = torch.Generator(device="cuda:0"); g.manual_seed(1)
    obs = torch.randn(n, 26, device="cuda:0", generator=g) * 5.0
    taus = torch.rand(n, 32, device="cuda:0", generator=g)
    a, q = fused_act(net, obs, 0.0, 1.0, taus=taus, want_qvals=True)
    assert a.dtype == torch.int32 and bool((a.long() == q.argmax(1)).all())
    g2 = torch.Generator(device="cuda:0"); g2.manual_seed(7)
    a1 = fused_act(net, obs, 1.0, 1.0, taus=taus, generator=g2)            # always explore
    cnt = torch.bincount(a1.long(), minlength=9).float() / n
    assert bool(((a1 >= 0) & (a1 < 9)).all()) and float((cnt - 1 / 9).abs().max()) < 0.01
    g3 = torch.Generator(device="cuda:0"); g3.manual_seed(8)
    a2 = fused_act(net, obs, 0.3, 1.0, taus=taus, generator=g3)
    frac_greedy = float((a2 == a).float().mean())                          # 0.7 + 0.3 * P(random == greedy)
    assert 0.70 < frac_greedy < 0.76


@pytest.mark.parametrize("fused_train", [True, False])
def test_g7_reference_vectors_on_device(torch, fused_train):
    """Golden vectors produced by the reference's own thirdparty/IQN code (tests/golden/make_golden.py, G7), on
    the GPU: (1) the fused MFMA act kernel reproduces the reference's Q-values for injected taus (seeded init and
    the shipped checkpoint), (2) one `IQNAgent.train` step on the device reproduces the reference's loss, clipped
    gradients and post-Adam parameters -- through the fused HIP step (csrc/iqn_train.hip) and through PyTorch autograd.
    Tolerances = the CPU test's (tests/test_iqn_cpu.py), f32 throughout."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.iqn.fused_act import fused_qvals
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    Z = np.load(os.path.join(G, "g7_iqn.npz"))
    dev = "cuda:0"
    obs = torch.from_numpy(Z["obs"]).to(dev); taus = torch.from_numpy(Z["taus32"]).to(dev)
    net = ObsEncoder(26, 9, seed=7, device=dev)
    for cvar in (1.0, 0.5):
        q = fused_qvals(net, obs, cvar, taus=taus).cpu().numpy()
        np.testing.assert_allclose(q, Z[f"qvals_cvar{cvar}"], rtol=1e-5, atol=1e-5)
    pre = ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), dev)
    q = fused_qvals(pre, obs, 1.0, taus=taus).cpu().numpy()
    np.testing.assert_allclose(q, Z["pretrained_quantiles"].mean(axis=1), rtol=1e-5, atol=1e-4)

    agent = IQNAgent(26, 9, BATCH_SIZE=16, seed=7, BUFFER_SIZE=64, device=dev)
    assert agent.use_fused_train          # the HIP step is the default on the GPU
    agent.use_fused_train = fused_train
    agent.qnetwork_target.load_state_dict({k[4:]: torch.from_numpy(Z[k]).to(dev) for k in Z.files if k.startswith("tgt_")})
    exp = tuple(torch.from_numpy(Z[k]).to(dev) for k in ("obs", "actions", "rewards", "next_obs", "dones"))
    loss = agent.train(exp, taus_target=torch.from_numpy(Z["taus8_target"]).to(dev),
                       taus_local=torch.from_numpy(Z["taus8_local"]).to(dev))
    np.testing.assert_allclose(float(loss), float(Z["train_loss"]), rtol=1e-5)
    for k, p in agent.qnetwork_local.named_parameters():
        np.testing.assert_allclose(p.grad.cpu().numpy(), Z["grad_" + k], rtol=1e-4, atol=1e-6, err_msg=k)
        np.testing.assert_allclose(p.detach().cpu().numpy(), Z["after_" + k], rtol=0, atol=2e-6, err_msg=k)


def _random_batch(torch, B, g):
    dev = "cuda:0"
    obs = torch.randn(B, 26, device=dev, generator=g) * 5
    obs[:, 4:] = torch.where(torch.rand(B, 22, device=dev, generator=g) < 0.5, torch.zeros((), device=dev), obs[:, 4:])
    return (obs, torch.randint(0, 9, (B, 1), device=dev, generator=g), torch.randn(B, 1, device=dev, generator=g) * 3,
            obs + 0.3 * torch.randn(B, 26, device=dev, generator=g), (torch.rand(B, 1, device=dev, generator=g) < 0.1).float())


@pytest.mark.parametrize("B", [2, 32, 256])
def test_fused_train_step_equals_pytorch(torch, B):
    """One optimizer step, same batch / taus / weights, fused HIP kernels vs PyTorch autograd + clip_grad_norm_ + Adam:
    loss to 1e-6 relative, clipped gradient to 1e-6 of its largest entry, post-Adam parameters to 2e-6 (float32
    rounding only: both are exact-f32 products with different summation orders).  A second step from the updated
    weights checks the Adam moments / bias correction (step counter) carried on the device."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    g = torch.Generator(device=dev); g.manual_seed(100 + B)
    a = IQNAgent(26, 9, BATCH_SIZE=B, seed=3, BUFFER_SIZE=1024, device=dev); a.use_fused_train = False
    b = IQNAgent(26, 9, BATCH_SIZE=B, seed=3, BUFFER_SIZE=1024, device=dev); b.use_fused_train = True
    with torch.no_grad():                       # a target network that differs from the local one
        for p in a.qnetwork_target.parameters():
            p.add_(0.05 * torch.randn(p.shape, device=dev, generator=g))
    b.qnetwork_target.load_state_dict(a.qnetwork_target.state_dict())
    for step in range(2):
        exp = _random_batch(torch, B, g)
        tt, tl = torch.rand(B, 8, device=dev, generator=g), torch.rand(B, 8, device=dev, generator=g)
        b.qnetwork_local.load_state_dict(a.qnetwork_local.state_dict())     # same starting weights for this step
        la, lb = float(a.train(exp, tt, tl)), float(b.train(exp, tt, tl))
        assert abs(la - lb) <= 1e-6 * abs(la), (la, lb)
        ga = torch.cat([p.grad.reshape(-1) for p in a.qnetwork_local.parameters()])
        gb = torch.cat([p.grad.reshape(-1) for p in b.qnetwork_local.parameters()])
        assert float((ga - gb).abs().max()) <= 1e-6 * float(ga.abs().max()), step
        assert abs(float(torch.linalg.vector_norm(gb)) - min(0.5, float(torch.linalg.vector_norm(gb)))) < 1e-6   # clipped
        if step == 0:      # identical Adam state only on the first step (b's weights are re-synchronised, its moments are its own)
            pa = torch.cat([p.detach().reshape(-1) for p in a.qnetwork_local.parameters()])
            pb = torch.cat([p.detach().reshape(-1) for p in b.qnetwork_local.parameters()])
            assert float((pa - pb).abs().max()) <= 2e-6      # first Adam step = lr * g / (|g| + eps): gradients of ~eps size amplify rounding
    assert int(b._fused.step_dev.item()) == 2 and b.grad_steps == 2


def test_fused_train_from_replay_ring(torch):
    """train_from_memory(): the HIP step gathers its batch from the ring by the sampled indices -- same result as
    handing it the gathered tensors; indices are distinct, in range, differ between calls; the parameters stay
    views of the flat buffers (checkpoint / soft_update / act kernel see the update)."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    g = torch.Generator(device=dev); g.manual_seed(5)
    B = 64
    a = IQNAgent(26, 9, BATCH_SIZE=B, seed=4, BUFFER_SIZE=500, device=dev)
    b = IQNAgent(26, 9, BATCH_SIZE=B, seed=4, BUFFER_SIZE=500, device=dev)
    for ag in (a, b):
        g.manual_seed(5)
        for _ in range(3):
            s_, ac, r, ns, d = _random_batch(torch, 150, g)
            ag.memory.add_batch(s_, ac.view(-1), r.view(-1), ns, d.view(-1))
    assert len(a.memory) == 450
    before = [p.detach().clone() for p in a.qnetwork_local.parameters()]
    la = a.train_from_memory()
    ft = a._fused
    idx, taus = ft._idx[B].clone(), ft._taus[B].clone()
    assert idx.unique().numel() == B and int(idx.min()) >= 0 and int(idx.max()) < 450
    m = b.memory
    lb = b.train((m.states[idx], m.actions[idx], m.rewards[idx], m.next_states[idx], m.dones[idx]), taus[0], taus[1])
    assert abs(float(la) - float(lb)) <= 1e-6 * abs(float(la))
    for p, q, p0 in zip(a.qnetwork_local.parameters(), b.qnetwork_local.parameters(), before):
        assert float((p.detach() - q.detach()).abs().max()) <= 2e-6 and float((p.detach() - p0).abs().max()) > 0
        assert p.data_ptr() >= ft.local.data_ptr() and p.data_ptr() < ft.local.data_ptr() + ft.local.numel() * 4
    a.train_from_memory()
    assert not torch.equal(ft._idx[B], idx) and not torch.equal(ft._taus[B], taus)
    a.soft_update(a.qnetwork_local, a.qnetwork_target)
    assert torch.equal(ft.target, ft.local)
    # sampling without replacement is uniform: inclusion frequency of every row -> B / size
    cnt = torch.zeros(450, device=dev)
    for _ in range(1500):
        i2, t2 = ft.sample(450, B)
        cnt[i2] += 1
    freq = (cnt / 1500).cpu().numpy()
    assert abs(freq.mean() - B / 450) < 1e-6 and freq.std() < 1.5 * np.sqrt(B / 450 * (1 - B / 450) / 1500)
    assert 0.49 < float(t2.mean()) < 0.51 and float(t2.min()) >= 0.0 and float(t2.max()) < 1.0


def test_fused_train_is_deterministic_and_seeded(torch):
    """No float atomics anywhere in the gradient step: two learners with the same seed produce BITWISE identical
    parameters, losses, sampled indices and taus over several train_from_memory() calls; a different seed samples
    differently."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    runs = []
    for seed in (11, 11, 12):
        g = torch.Generator(device=dev); g.manual_seed(77)
        ag = IQNAgent(26, 9, BATCH_SIZE=256, seed=seed, BUFFER_SIZE=2000, device=dev)
        s_, ac, r, ns, d = _random_batch(torch, 2000, g)
        ag.memory.add_batch(s_, ac.view(-1), r.view(-1), ns, d.view(-1))
        losses = [float(ag.train_from_memory()) for _ in range(6)]
        ft = ag._fused
        runs.append((losses, ft.local.clone(), ft._idx[256].clone(), ft._taus[256].clone()))
    (l0, p0, i0, t0), (l1, p1, i1, t1), (l2, p2, i2, t2) = runs
    assert l0 == l1 and torch.equal(p0, p1) and torch.equal(i0, i1) and torch.equal(t0, t1)
    assert not torch.equal(i0, i2) and not torch.equal(t0, t2)
    assert all(np.isfinite(l0)) and bool(torch.isfinite(p0).all())


def test_iqn_c_abi_argument_checks(torch):
    """Error behaviour of the IQN entry points: integer status codes, nothing launched on bad arguments."""
    import ctypes as C
    from distributional_rl_navigation_amd import _capi
    L = _capi.lib()
    # 128 partial rows of 35 788 floats + 128 loss partials + 280 norm partials + 128 x 16 8-byte hand-off granules + epoch / tickets /
    # staging tag / magic word / status word (16 words) + the 280 tagged norm partials of the fused reduction + Adam launch; in brackets the fused step's 128 8-byte
    # row-complete words, the 128 buddy words, 8 x 64 XCD-local row-complete words, 128 tagged loss partials, 128 "which XCD" words and the eight XCD group rows as 8-byte
    # granules; + the staged next batch (256 slots of 72 floats)
    assert L.mn_iqn_train_workspace_floats(256) == 128 * 35788 + 128 + 280 + 2 * 128 * 16 + 16 + 2 * 280 + (2 * 128 + 128 + 512 + 2 * 128 + 2 * 128 + 16 * 35788) + 256 * 72
    assert L.mn_iqn_train_workspace_floats(255) == -1 and L.mn_iqn_train_workspace_floats(0) == -1
    dev = "cuda:0"
    st = torch.zeros(2, dtype=torch.int64, device=dev); idx = torch.zeros(2048, dtype=torch.int64, device=dev)
    taus = torch.zeros(64, device=dev)
    p = lambda t: C.c_void_p(t.data_ptr())
    INVALID = -1
    assert L.mn_iqn_sample(100, 256, p(st), p(idx), p(taus), 64, None) == INVALID       # ring smaller than the batch
    assert L.mn_iqn_sample(10000, 2048, p(st), p(idx), p(taus), 64, None) == INVALID    # batch > 1024
    assert L.mn_iqn_sample(10000, 16, None, p(idx), p(taus), 64, None) == INVALID
    assert L.mn_iqn_sample(10000, 16, p(st), p(idx), None, 0, None) == 0                # taus optional
    torch.cuda.synchronize()
    assert int(st[1]) == 1 and idx[:16].unique().numel() == 16
    f = torch.zeros(35785, device=dev); ws = torch.zeros(L.mn_iqn_train_workspace_floats(2), device=dev)
    ring = (torch.zeros(4, 26, device=dev), torch.zeros(4, 26, device=dev), torch.zeros(4, 1, dtype=torch.int64, device=dev),
            torch.zeros(4, 1, device=dev), torch.zeros(4, 1, device=dev))
    args = lambda B, K: (p(ring[0]), p(ring[1]), p(ring[2]), p(ring[3]), p(ring[4]), p(idx), p(taus), p(taus), p(f), p(f), p(ws),
                         p(f), p(taus), B, K, C.c_float(0.99), None)
    assert L.mn_iqn_train_grad(*args(3, 8)) == INVALID      # odd batch
    assert L.mn_iqn_train_grad(*args(2, 32)) == INVALID     # training uses 8 taus
    assert L.mn_iqn_train_adam(p(f), p(f), p(f), p(f), None, p(ws), 2, 1e-4, 0.9, 0.999, 1e-8, 0.5, C.c_float(1.0), 0, None) == INVALID
    assert L.mn_iqn_train_adam(p(f), p(f), p(f), p(f), p(st), p(ws), 2, 1e-4, 0.9, 0.999, 1e-8, 0.5, C.c_float(0.0), 0, None) == INVALID   # grad_scale must be positive
    assert L.mn_iqn_train_set_mode(2) == INVALID and L.mn_iqn_train_set_mode(0) == 0
    assert L.mn_iqn_train_workspace_init(None, 2, None) == INVALID and L.mn_iqn_train_workspace_init(p(ws), 3, None) == INVALID
    # a workspace that was never initialised is refused on the device: NaN loss, parameters and moments untouched
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    ag = IQNAgent(26, 9, BATCH_SIZE=32, BUFFER_SIZE=256, device=dev, seed=3)
    ag.memory.add_batch(*_random_batch(torch, 200, torch.Generator(device=dev).manual_seed(4)))
    ft = ag._fused_trainer()
    assert np.isfinite(float(ag.train_from_memory()))
    garbage = torch.full_like(ft._ws, 3.0e9)
    ft._ws_by_batch[32] = ft._ws = garbage
    before, m_before, step_before = ft.local.clone(), ft.exp_avg.clone(), int(ft.step_dev)
    assert np.isnan(float(ag.train_from_memory()))
    assert torch.equal(ft.local, before) and torch.equal(ft.exp_avg, m_before) and int(ft.step_dev) == step_before
    assert L.mn_iqn_train_workspace_init(p(garbage), 32, None) == 0
    assert np.isfinite(float(ag.train_from_memory())) and not torch.equal(ft.local, before)
    assert L.mn_iqn_act(None, p(ring[0]), p(taus), None, None, None, C.c_float(0.0), p(idx), None, 4, 32, None) == INVALID


def test_act_with_library_drawn_taus(torch):
    """`mn_iqn_act_rng`: the launch that packs the weights also draws the call's taus and exploration uniforms.
    (1) Feeding the draws it left in the scratch buffer back through the injected-taus path reproduces Q-values and
    actions bitwise; (2) taus are U[0,1) * cvar (scalar and per-row), fresh on every call, reproducible from the seed;
    (3) exploration happens with probability eps and is uniform over the 9 actions."""
    from distributional_rl_navigation_amd.iqn.fused_act import ActRng, fused_act, fused_qvals
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    dev = "cuda:0"
    net = ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), dev)
    n = 30001
    g = torch.Generator(device=dev); g.manual_seed(3)
    obs = torch.randn(n, 26, device=dev, generator=g) * 5.0
    rng = ActRng(123, dev)
    a, q = fused_act(net, obs, 0.0, 1.0, rng=rng, want_qvals=True)
    d = rng.draws(n, 32).clone()
    taus, u = d[:n * 32].view(n, 32), d[n * 32:]
    assert int(rng.state[1]) == 1
    assert float(taus.min()) >= 0.0 and float(taus.max()) < 1.0 and abs(float(taus.mean()) - 0.5) < 2e-3
    assert abs(float(taus.var()) - 1 / 12) < 2e-3 and abs(float(u.mean()) - 0.5) < 1e-2
    assert abs(float(torch.corrcoef(torch.stack((taus[:, 0], taus[:, 1])))[0, 1])) < 0.02       # neighbouring draws uncorrelated
    a2, q2 = fused_act(net, obs, 0.0, 1.0, taus=taus, want_qvals=True)
    assert torch.equal(q, q2) and torch.equal(a, a2)
    # fresh draws on the next call; same seed -> same sequence
    fused_act(net, obs, 0.0, 1.0, rng=rng)
    assert not torch.equal(rng.draws(n, 32), d) and int(rng.state[1]) == 2
    rng2 = ActRng(123, dev)
    fused_act(net, obs, 0.0, 1.0, rng=rng2)
    assert torch.equal(rng2.draws(n, 32), d)
    # cvar scaling: scalar and per-row
    fused_act(net, obs, 0.0, 0.25, rng=rng)
    t = rng.draws(n, 32)[:n * 32]
    assert float(t.max()) < 0.25 and abs(float(t.mean()) - 0.125) < 1e-3
    cv = torch.rand(n, device=dev, generator=g)
    fused_act(net, obs, 0.0, cv, rng=rng)
    t = rng.draws(n, 32)[:n * 32].view(n, 32)
    assert bool((t <= cv.view(-1, 1)).all()) and abs(float((t / cv.view(-1, 1)).mean()) - 0.5) < 2e-3
    # epsilon-greedy from the library's own uniforms
    greedy = fused_act(net, obs, 0.0, 1.0, taus=taus)
    acts = fused_act(net, obs, 0.3, 1.0, rng=rng)
    u = rng.draws(n, 32)[n * 32:]
    explored = ~(u > 0.3)
    assert abs(float(explored.float().mean()) - 0.3) < 0.01
    hist = torch.bincount(acts[explored].long(), minlength=9).float()
    assert float((hist / hist.sum() - 1 / 9).abs().max()) < 0.01


def test_hip_and_torch_gradient_steps_share_one_adam_state(torch):
    """`use_fused_train` may be flipped mid-run: the Adam moments are ONE set of buffers (torch.optim.Adam's state
    tensors are views of the HIP step's flat moments) and the step count is handed over, so a run that alternates
    between the two paths equals -- to float32 rounding -- a run that stays on either."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    g = torch.Generator(device=dev); g.manual_seed(11)
    batches = [_random_batch(torch, 64, g) for _ in range(6)]
    taus = [(torch.rand(64, 8, device=dev, generator=g), torch.rand(64, 8, device=dev, generator=g)) for _ in range(6)]

    def run(pattern):
        ag = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=128, device=dev, seed=4)
        for use_hip, b, (tt, tl) in zip(pattern, batches, taus):
            ag.use_fused_train = use_hip
            ag.train(b, taus_target=tt, taus_local=tl)
        return ag

    hip = run([True] * 6)
    mixed = run([True, True, False, False, True, False])
    tor = run([False] * 6)
    flat = lambda ag: torch.cat([p.detach().reshape(-1) for p in ag.qnetwork_local.parameters()])
    # 6 steps of lr 1e-4: a restarted optimizer (bias correction back at t = 1, moments at zero) would move the weights
    # by ~1e-4 per step relative to the continued one; the shared state keeps all three runs within rounding
    assert float((flat(hip) - flat(tor)).abs().max()) < 5e-6
    assert float((flat(mixed) - flat(hip)).abs().max()) < 5e-6
    p0 = next(iter(mixed.qnetwork_local.parameters()))
    assert int(float(mixed.optimizer.state[p0]["step"])) == 6                    # torch's counter carries all 6 steps
    mixed.use_fused_train = True
    mixed.train(batches[0], taus_target=taus[0][0], taus_local=taus[0][1])
    assert int(mixed._fused.step_dev) == 7
    st = mixed.optimizer.state[p0]
    assert st["exp_avg"].data_ptr() == mixed._fused.exp_avg.data_ptr()           # same memory, not a copy


def test_sampled_gradient_step_equals_sample_then_step(torch):
    """`mn_iqn_train_grad_sampled` draws the batch inside the forward / backward kernel (every workgroup runs the draw and keeps its
    own rows): same rows, same taus, same gradient step, bit for bit, as `mn_iqn_sample` followed by `mn_iqn_train_grad` from the
    same generator state, and the call counter advances by one per step either way."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    agents = []
    for _ in range(2):
        g = torch.Generator(device=dev); g.manual_seed(21)
        ag = IQNAgent(26, 9, BATCH_SIZE=256, seed=9, BUFFER_SIZE=3000, device=dev)
        s_, ac, r, ns, d = _random_batch(torch, 2500, g)
        ag.memory.add_batch(s_, ac.view(-1), r.view(-1), ns, d.view(-1))
        agents.append(ag)
    a, b = agents
    for step in range(4):
        la = a.train_from_memory()                                             # sampled inside the kernel
        m, ft = b.memory, b._fused_trainer()
        b._enter_train_path("hip")
        idx, taus = ft.sample(m.size, 256)                                     # two-launch form
        lb = ft.step((m.states, m.actions, m.rewards, m.next_states, m.dones), idx, taus[0], taus[1])
        fa = a._fused
        assert torch.equal(fa._idx[256], idx) and torch.equal(fa._taus[256], taus), step
        assert float(la) == float(lb) and torch.equal(fa.local, ft.local) and torch.equal(fa.grad, ft.grad), step
        assert torch.equal(fa.rng_state, ft.rng_state) and int(fa.rng_state[1]) == step + 1
        assert idx.unique().numel() == 256 and int(idx.max()) < 2500
    # a small ring and a batch that is not a multiple of the workgroup count's granularity
    c = IQNAgent(26, 9, BATCH_SIZE=32, seed=3, BUFFER_SIZE=64, device=dev)
    g = torch.Generator(device=dev); g.manual_seed(2)
    s_, ac, r, ns, d = _random_batch(torch, 40, g)
    c.memory.add_batch(s_, ac.view(-1), r.view(-1), ns, d.view(-1))
    for _ in range(3):
        assert np.isfinite(float(c.train_from_memory()))
        i3 = c._fused._idx[32]
        assert i3.unique().numel() == 32 and int(i3.min()) >= 0 and int(i3.max()) < 40


def _filled_agent(torch, seed, B=256, ring=3000, fill=2500):
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    g = torch.Generator(device=dev); g.manual_seed(21)
    ag = IQNAgent(26, 9, BATCH_SIZE=B, seed=seed, BUFFER_SIZE=ring, device=dev)
    s_, ac, r, ns, d = _random_batch(torch, fill, g)
    ag.memory.add_batch(s_, ac.view(-1), r.view(-1), ns, d.view(-1))
    with torch.no_grad():                       # a target network that differs from the local one
        for p in ag.qnetwork_target.parameters():
            p.add_(0.05 * torch.randn(p.shape, device=dev, generator=g))
    return ag


def test_two_role_launch_equals_local_only_launch_bitwise(torch):
    """Round-3 forward / backward launch: TARGET workgroups hand their 16 TD targets to the LOCAL workgroup of the same two batch
    elements inside the launch (self-tagged granules).  `mn_iqn_train_set_mode(1)` makes every local workgroup run the target
    forward itself -- no inter-workgroup communication.  Same arithmetic: losses, gradients, parameters, drawn batches are
    bit-identical over several steps (batch 256 = 128 + 128 workgroups, and a small batch)."""
    from distributional_rl_navigation_amd import _capi
    L = _capi.lib()
    for B, ring, fill in ((256, 3000, 2500), (6, 64, 40)):
        runs = []
        for mode in (0, 1):
            assert L.mn_iqn_train_set_mode(mode) == 0
            try:
                ag = _filled_agent(torch, 9, B, ring, fill)
                losses = [float(ag.train_from_memory()) for _ in range(5)]
                ft = ag._fused
                runs.append((losses, ft.local.clone(), ft.grad.clone(), ft._idx[B].clone(), ft._taus[B].clone(), int(ft.step_dev)))
            finally:
                L.mn_iqn_train_set_mode(0)
        (l0, p0, g0, i0, t0, s0), (l1, p1, g1, i1, t1, s1) = runs
        assert l0 == l1 and torch.equal(p0, p1) and torch.equal(g0, g1) and torch.equal(i0, i1) and torch.equal(t0, t1)
        assert s0 == s1 == 5 and all(np.isfinite(l0))


def test_gradient_step_under_a_busy_gpu_is_bitwise_the_quiet_step(torch):
    """The in-launch hand-off, the Adam step-counter ticket and the reduction must not depend on dispatch timing: the same 40
    gradient steps with the act kernel of 65 536 envs running on a second stream (it holds every CU, so the gradient step's
    workgroups are dispatched late and unevenly) give bit-identical parameters, and the step counter counts every step."""
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    dev = "cuda:0"
    quiet = _filled_agent(torch, 5)
    for _ in range(40):
        quiet.train_from_memory()
    busy = _filled_agent(torch, 5)
    obs = torch.randn(65536, 26, device=dev) * 5.0
    actor = _filled_agent(torch, 6, B=32, ring=64, fill=40)
    side = torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for it in range(40):
        if it % 4 == 0:
            with torch.cuda.stream(side):
                fused_act(actor.qnetwork_local, obs, 0.1, 1.0, generator=None)
        busy.train_from_memory()
    torch.cuda.synchronize()
    assert torch.equal(quiet._fused.local, busy._fused.local) and torch.equal(quiet._fused.exp_avg_sq, busy._fused.exp_avg_sq)
    assert int(quiet._fused.step_dev) == int(busy._fused.step_dev) == 40


def test_permutation_sampler_properties(torch):
    """ReplayBuffer.sample = random.sample(memory, k): k distinct uniform rows.  The kernels read slot k's row from a keyed
    pseudo-random permutation of [0, n): distinct for every ring size (powers of two, just above / below, tiny, the 100 000 of
    the headline configuration), uniform (inclusion frequency of every row), no first-slot bias, fresh per call."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    ag = IQNAgent(26, 9, BATCH_SIZE=64, seed=4, BUFFER_SIZE=128, device=dev)
    ft = ag._fused_trainer()
    for n in (64, 65, 127, 128, 129, 1000, 4096, 4097, 100_000, 2 ** 31 - 1):
        seen = set()
        for _ in range(3):
            idx, _t = ft.sample(n, 64)
            v = idx.cpu().numpy()
            assert len(np.unique(v)) == 64 and v.min() >= 0 and v.max() < n, n
            seen.add(tuple(v))
        assert len(seen) == 3
    n = 300
    cnt = np.zeros(n); first = np.zeros(n)
    reps = 3000
    for _ in range(reps):
        v = ft.sample(n, 64)[0].cpu().numpy()
        cnt[v] += 1; first[v[0]] += 1
    p = 64 / n
    assert abs(cnt.mean() / reps - p) < 1e-9 and (cnt / reps).std() < 1.5 * np.sqrt(p * (1 - p) / reps)
    assert first.max() < 40 and (first > 0).sum() > 0.95 * n          # slot 0 is uniform over the ring as well (mean 10 per row)


def test_staged_batch_step_equals_drawn_batch_step_bitwise(torch):
    """`train_from_memory()` lets the reduction kernel of step k stage step k + 1's batch (rows, transitions, taus) and starts
    step k + 1 from it (MN_TRAIN_STAGE_NEXT / MN_TRAIN_USE_STAGED).  Same batch, same arithmetic: bit-identical to steps that draw
    and gather inside the launch -- also across a write to the ring (the staged batch is then not used: the ring version moved),
    a change of the ring size and a generator state set from outside (tag mismatch on the device)."""
    a = _filled_agent(torch, 9)
    b = _filled_agent(torch, 9)
    g = torch.Generator(device="cuda:0"); g.manual_seed(77)
    extra = _random_batch(torch, 200, g)

    def step_b():
        m, ft = b.memory, b._fused_trainer()
        b._enter_train_path("hip")
        return ft.step_sampled((m.states, m.actions, m.rewards, m.next_states, m.dones), m.size, b.BATCH_SIZE)      # no staging

    for it in range(12):
        if it == 5:      # ring written between two steps (and it grows: 2500 -> 2700 rows)
            for ag in (a, b):
                ag.memory.add_batch(extra[0], extra[1].view(-1), extra[2].view(-1), extra[3], extra[4].view(-1))
        if it == 8:
            for ag in (a, b):
                ag._fused.rng_state.copy_(torch.tensor([int(ag._fused.rng_state[0]), 1000], dtype=torch.int64))
        la, lb = float(a.train_from_memory()), float(step_b())
        assert la == lb, it
        assert torch.equal(a._fused.local, b._fused.local) and torch.equal(a._fused._idx[256], b._fused._idx[256]), it
        assert torch.equal(a._fused._taus[256], b._fused._taus[256]) and torch.equal(a._fused.rng_state, b._fused.rng_state), it
    assert a._fused._staged_key is not None and b._fused._staged_key is None


def test_n_step_agent_runs_the_vector_loop(torch):
    """IQNAgent(n_step = 3) on the HIP vector env: the loop takes the step + add_vector_step path (the fused append stores 1-step
    transitions), the ring receives one 3-step transition per env and vector step once the windows are full, the gradient step
    discounts with gamma^3."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(256, seed=0, device="cuda:0", precision="f64")
    ag = IQNAgent(26, 9, n_step=3, BATCH_SIZE=64, BUFFER_SIZE=4096, device="cuda:0", seed=1, learning_starts=0, UPDATE_EVERY=2)
    stats = ag.learn_vec(total_vector_steps=8, train_env=env, verbose=False)
    torch.cuda.synchronize()
    assert len(ag.memory) == 256 * (8 - 2) and ag.grad_steps >= 2 and np.isfinite(float(stats["loss"]))
    env.close()


def test_one_and_two_launch_steps_equal_the_three_launch_step_bitwise(torch):
    """`mn_iqn_train_step` (round 4): the reduction, clip and Adam as ONE launch in which every block reduces its own parameters' partial
    gradients and exchanges the norm partials as self-tagged granules (two launches per step), or as a third workgroup role of the forward /
    backward launch itself (MN_TRAIN_ONE_LAUNCH: one launch per step) -- against `mn_iqn_train_grad*` + `mn_iqn_train_adam` (three launches):
    losses, clipped gradients, parameters, moments, Adam step, generator state bit-identical over sampled steps (staged batches incl.), a ring
    write in between, given-batch steps with injected taus (batch 64: 32 rows, 4 per XCD group), a captured 8-step hipGraph, and in the local-only
    workgroup mode (in which the fused step does not exist: the library takes two launches).  The fused step also with workgroups that
    pretend to have landed on another XCD (every fifth / all / all of one group: their rows go through memory, they take no share of the group sum)."""
    from distributional_rl_navigation_amd import _capi
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    runs = []
    # (whole step in the library, fused step asked for, pretended XCD misplacement 0..3, local-only workgroup mode)
    cases = ((True, True, 0, 0), (True, False, 0, 0), (False, False, 0, 0), (True, True, 0, 1), (True, True, 1, 0), (True, True, 2, 0), (True, True, 3, 0))
    for two, one, misplace, mode in cases:
        ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=2048, device=dev, seed=11)
        ag.two_launch_step, ag.one_launch_step, ag._test_misplace = two, one, misplace
        _capi.lib().mn_iqn_train_set_mode(mode)      # 1: every workgroup computes its own TD targets, no target role in the launch
        g = torch.Generator(device=dev); g.manual_seed(5)
        ag.memory.add_batch(*_random_batch(torch, 2048, g))
        losses = [float(ag.train_from_memory()) for _ in range(12)]
        ag.memory.add_batch(*_random_batch(torch, 300, g))
        losses += [float(ag.train_from_memory()) for _ in range(5)]
        for k in range(3):
            exp = _random_batch(torch, 64, g)
            tt = torch.rand(64, 8, device=dev, generator=g); tl = torch.rand(64, 8, device=dev, generator=g)
            losses.append(float(ag.train(exp, taus_target=tt, taus_local=tl)))
        ag.use_fused_graph = True
        losses.append(float(ag.train_steps_from_memory(8)))
        losses.append(float(ag.train_steps_from_memory(8)))
        ft = ag._fused
        assert ft._two_launches() == two and ft.timeouts() == 0
        assert ft.launches_per_step(256) == (3 if not two else (1 if one and mode == 0 else 2))
        runs.append((losses, ft.local.clone(), ft.grad.clone(), ft.exp_avg.clone(), ft.exp_avg_sq.clone(), int(ft.step_dev), ft.rng_state.clone(), ag.grad_steps))
    _capi.lib().mn_iqn_train_set_mode(0)
    ref = runs[2]      # three launches
    assert all(np.isfinite(ref[0]))
    for r in runs[:2] + runs[3:]:
        assert r[0] == ref[0]
        for x, y in zip(r[1:5], ref[1:5]):
            assert torch.equal(x, y)
        assert r[5] == ref[5] == 12 + 5 + 3 + 16 and torch.equal(r[6], ref[6]) and r[7] == ref[7]


def test_launch_plan_follows_the_device_size_and_small_devices_fall_back_bitwise(torch):
    """The fused forms wait inside a launch for other workgroups of the same launch, so the library plans them from the device's CU count, not from the constant
    256: `mn_iqn_train_set_cu_limit` pretends a smaller device.  Batch 256: the whole MI355X -> one launch (256 workgroups, one CU each); 200 or 100 CUs ->
    two (the fused step's workgroups would not all be resident together; the 140 blocks of the reduction + Adam launch are); 8 -> three launches, nothing waits for a
    sibling.  All bit-identical, no bounded wait ran out."""
    from distributional_rl_navigation_amd import _capi
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev, L = "cuda:0", _capi.lib()
    runs = []
    try:
        for limit, want in ((0, 1), (200, 2), (100, 2), (8, 3)):
            assert L.mn_iqn_train_set_cu_limit(limit) == 0
            ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=2048, device=dev, seed=11)
            g = torch.Generator(device=dev); g.manual_seed(5)
            ag.memory.add_batch(*_random_batch(torch, 2048, g))
            losses = [float(ag.train_from_memory()) for _ in range(10)]
            ft = ag._fused
            assert ft.launches_per_step(256) == want and L.mn_iqn_train_plan(256, 4, 1) == (want if want < 3 else 4)
            assert ft.timeouts() == 0
            runs.append((losses, ft.local.clone(), ft.grad.clone(), ft.exp_avg_sq.clone(), int(ft.step_dev), ft.rng_state.clone()))
    finally:
        L.mn_iqn_train_set_cu_limit(0)
    assert L.mn_iqn_train_set_cu_limit(-1) != 0 and L.mn_iqn_train_plan(255, 4, 0) < 0 and L.mn_iqn_train_plan(256, 4, 0) == 1
    a = runs[0]
    assert all(np.isfinite(a[0]))
    for b in runs[1:]:
        assert a[0] == b[0] and a[4] == b[4] == 10
        for x, y in ((a[1], b[1]), (a[2], b[2]), (a[3], b[3]), (a[5], b[5])):
            assert torch.equal(x, y)


@pytest.mark.parametrize("batch,misplace", [(256, 0), (256, 1), (64, 2), (100, 0)])
def test_training_events_fused_equals_three_launches_bitwise(torch, batch, misplace):
    """Training events as the batched loop runs them (`train_steps_from_memory`: G gradient steps back to back, the first from the batch the previous event's last
    step staged -- used while the ring has not moved, refused after ring writes): the fused one-launch step against the three-launch path over events of
    16 / 16 / 5 / 1 / 16 / 3 steps with ring writes in between -- every event's last loss, the parameters, both moments, the last clipped gradient, the Adam step
    and the generator state bit-identical, also with workgroups pretending to sit on another XCD and at batch 100 (no fused form: two launches).  No bounded wait ran out."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    runs = []
    for form in ("single", "three"):
        ag = IQNAgent(26, 9, BATCH_SIZE=batch, BUFFER_SIZE=2048, device=dev, seed=11)
        ag.two_launch_step, ag.one_launch_step, ag._test_misplace = form != "three", True, misplace
        g = torch.Generator(device=dev); g.manual_seed(5)
        ag.memory.add_batch(*_random_batch(torch, 2048, g))
        losses = []
        for ev, G in enumerate((16, 16, 5, 1, 16, 3)):
            if ev in (2, 4):
                ag.memory.add_batch(*_random_batch(torch, 300, g))      # the ring moves: the batch the previous event staged must not be used
            losses.append(float(ag.train_steps_from_memory(G)))
        ft = ag._fused
        assert ft.timeouts() == 0
        runs.append((losses, ft.local.clone(), ft.grad.clone(), ft.exp_avg.clone(), ft.exp_avg_sq.clone(), int(ft.step_dev), ft.rng_state.clone(), ag.grad_steps))
    single, three = runs
    assert all(np.isfinite(single[0])) and single[0] == three[0]
    for x, y in zip(single[1:5], three[1:5]):
        assert torch.equal(x, y)
    assert single[5] == three[5] == 57 and torch.equal(single[6], three[6]) and single[7] == three[7] == 57


@pytest.mark.parametrize("one_launch", [True, False])
def test_workspace_regions_next_to_the_norm_partials_survive_a_step(torch, one_launch):
    """Sentinel check (the round-5 miscompile wrote norm partials of virtual blocks that do not exist -- index >= 280 -- over the TD-target granules that
    follow them in the workspace): after every gradient step each of the 128 x 16 TD granules must still carry the step's tag (high word: equal across granules, non-zero,
    different from the previous step's) and a finite value."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    ag = IQNAgent(26, 9, BATCH_SIZE=256, BUFFER_SIZE=4096, device=dev, seed=2)
    ag.one_launch_step, ag.two_launch_step = one_launch, True
    g = torch.Generator(device=dev); g.manual_seed(1)
    ag.memory.add_batch(*_random_batch(torch, 4096, g))
    n_part, P_PAD, N_RED = 128, 35788, 280
    tdq = n_part * P_PAD + n_part + N_RED      # floats: partial rows | loss partials (128, a multiple of 4) | norm partials (280, a multiple of 4) | TD granules (u64 {value, tag})
    last = None
    for _ in range(6):
        assert np.isfinite(float(ag.train_from_memory()))
        torch.cuda.synchronize()
        ws = ag._fused._ws
        gr = ws[tdq:tdq + 2 * n_part * 16].view(torch.int32).view(n_part * 16, 2)      # little endian: [value bits, tag]
        tags = gr[:, 1]
        assert int(tags.min()) == int(tags.max()) != 0, (int(tags.min()), int(tags.max()))
        assert last is None or int(tags[0]) != last
        last = int(tags[0])
        assert bool(torch.isfinite(gr[:, 0].contiguous().view(torch.float32)).all())
    assert ag._fused.launches_per_step(256) == (1 if one_launch else 2) and ag._fused.timeouts() == 0


@pytest.mark.parametrize("batch", [16, 48, 100, 128, 384, 512, 1024])
def test_one_launch_step_at_other_batch_sizes(torch, batch):
    """The fused step away from batch 256: 16 (one row per XCD group; 62 workgroups that only run reduction + Adam blocks behind the 16 forward / backward ones),
    48 (three rows per group), 128 (eight; the 64 target workgroups run two reduction + Adam blocks each and 6 extra workgroups the rest), and the batches for
    which the library takes two launches -- 100 (its half is no multiple of 8), 384 / 512 / 1024 (more workgroups than CUs) -- bit-identical to the three-launch
    path over sampled steps incl. staged batches, and no workgroup away from its group's XCD."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    dev = "cuda:0"
    runs = []
    for one in (True, False):
        ag = IQNAgent(26, 9, BATCH_SIZE=batch, BUFFER_SIZE=4096, device=dev, seed=3)
        ag.two_launch_step, ag.one_launch_step = one, one
        g = torch.Generator(device=dev); g.manual_seed(9)
        ag.memory.add_batch(*_random_batch(torch, 4096, g))
        losses = [float(ag.train_from_memory()) for _ in range(10)]
        ft = ag._fused
        if one:
            assert ft.xcd_misplaced(batch) == 0 and ft.timeouts() == 0
            assert ft.launches_per_step(batch) == (1 if batch in (16, 48, 128) else 2)
        runs.append((losses, ft.local.clone(), ft.grad.clone(), ft.exp_avg_sq.clone(), int(ft.step_dev), ft.rng_state.clone()))
    a, b = runs
    assert all(np.isfinite(a[0])) and a[0] == b[0] and a[4] == b[4] == 10
    for x, y in ((a[1], b[1]), (a[2], b[2]), (a[3], b[3]), (a[5], b[5])):
        assert torch.equal(x, y)
