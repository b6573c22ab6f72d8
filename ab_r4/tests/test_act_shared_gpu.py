"""Launch-shared quantile fractions (`mn_iqn_set_tau_mode(ctx, 1)`, `fused_act(..., shared_taus=True)`; round 4).

When every row of an act launch is evaluated at the SAME 32 taus, layer 1 of the network -- relu(W1 cos(pi k tau) + b1),
thirdparty/IQN/model.py:141-157,176-178 -- is a [32 x 208] constant of the launch; the shared-tau kernel
(csrc/iqn_act_split.h: stage_sh, iqn_shared_prep_kernel) computes it once and runs only the Hadamard product, layers 2-3 and the
output stage per row.  Claim under test: for GIVEN taus a row's result is the same function as in the per-row kernels and in
the PyTorch `ObsEncoder` (float32 rounding apart), incl. act_eval's quantile output, the range-scaling stress cases and the
in-library draw path; the mode is opt-in and switching it leaves the per-row path untouched."""
import copy
import ctypes as C
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
DEV = "cuda:0"


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("needs a GPU")
    return t


def _net(torch, which):
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    return ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), DEV) if which == "pretrained" else ObsEncoder(26, 9, seed=7, device=DEV)


def _inputs(torch, n, scale, seed=11):
    g = torch.Generator(device=DEV); g.manual_seed(seed)
    obs = torch.randn(n, 26, device=DEV, generator=g) * scale
    obs[:, 4:][torch.rand(n, 22, device=DEV, generator=g) < 0.4] = 0.0      # sonar misses are exact zeros
    return obs, torch.rand(32, device=DEV, generator=g)


@pytest.mark.parametrize("which", ["seeded", "pretrained"])
@pytest.mark.parametrize("cvar", [1.0, 0.5])
def test_same_taus_for_every_row_of_g7_through_both_kernels_and_pytorch(torch, which, cvar):
    """G7's observations (the batch the reference's own forward was recorded on), the same 32 taus for every row: the per-row
    kernel fed the broadcast [n, 32] tensor, the shared-tau kernel fed the [32] row, and the PyTorch ObsEncoder -- 1e-5, the
    bound G7's reference Q-values are held to (tests/test_iqn_gpu.py)."""
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    Z = np.load(os.path.join(G, "g7_iqn.npz"))
    obs = torch.from_numpy(Z["obs"]).to(DEV)
    n = obs.shape[0]
    row = torch.from_numpy(Z["taus32"][0]).to(DEV)
    net = _net(torch, which)
    bc = row.view(1, 32).expand(n, 32).contiguous()
    with torch.no_grad():
        ref = net.get_qvals(obs, cvar, taus=bc)
    a0, q0 = fused_act(net, obs, 0.0, cvar, taus=bc, want_qvals=True)
    a1, q1 = fused_act(net, obs, 0.0, cvar, taus=row, want_qvals=True, shared_taus=True)
    tol = dict(rtol=1e-5, atol=1e-5 if which == "seeded" else 1e-4)      # (pretrained |Q| ~ 100: G7's own bound for that net)
    np.testing.assert_allclose(q1.cpu().numpy(), ref.cpu().numpy(), **tol)
    np.testing.assert_allclose(q1.cpu().numpy(), q0.cpu().numpy(), **tol)
    assert bool((a1.long() == q1.argmax(1)).all())
    # the reference's own recorded Q-values are for per-row taus; row 0 of that batch IS evaluated at `row`
    if cvar in (1.0, 0.5) and which == "seeded":
        np.testing.assert_allclose(q1[0].cpu().numpy(), Z[f"qvals_cvar{cvar}"][0], rtol=1e-5, atol=1e-5)


@pytest.mark.parametrize("which", ["seeded", "pretrained"])
def test_shared_tau_kernel_is_float32_class_against_float64(torch, which):
    """Error against a FLOAT64 evaluation of the network, with the exact-f32 MFMA kernel as the yardstick (the bar of
    tests/test_act_split_gpu.py)."""
    from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
    net = _net(torch, which)
    obs, row = _inputs(torch, 16384, 5.0)
    n = obs.shape[0]
    bc = row.view(1, 32).expand(n, 32).contiguous()
    with torch.no_grad():
        ref = copy.deepcopy(net).double().get_qvals(obs.double(), 1.0, taus=bc.double())
    ctx = act_context(net)
    try:
        ctx.set_variant(0)
        _, qe = fused_act(net, obs, 0.0, 1.0, taus=bc, want_qvals=True)
    finally:
        ctx.set_variant(ctx.DEFAULT_VARIANT)
    a, qs = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)

    def err(q):
        d = (q.double() - ref).abs()
        return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())
    (mx_e, rms_e), (mx_s, rms_s) = err(qe), err(qs)
    assert rms_s < 1.25 * rms_e + 2e-8 and mx_s < 1.5 * mx_e + 1e-7, (mx_e, rms_e, mx_s, rms_s)
    assert rms_s < 1e-6 and mx_s < 3e-6
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5 * float(ref.abs().max())
    assert torch.equal(a.long()[clear], ref.argmax(dim=1)[clear])


@pytest.mark.parametrize("case", ["obs x 1e6", "obs x 1e-6", "obs zero", "weights x 30", "weights x 1e-3", "one huge weight", "one env huge among small"])
def test_range_scaling_cases(torch, case):
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    net = _net(torch, "seeded")
    obs, row = _inputs(torch, 4096, 5.0)
    with torch.no_grad():
        if case == "obs x 1e6": obs *= 1e6
        elif case == "obs x 1e-6": obs *= 1e-6
        elif case == "obs zero": obs.zero_()
        elif case == "weights x 30":
            for p in net.parameters():
                if p.dim() == 2: p.mul_(30.0)
        elif case == "weights x 1e-3":
            for p in net.parameters():
                if p.dim() == 2: p.mul_(1e-3)
        elif case == "one huge weight":
            net.hidden_layer.weight[3, 100] = 500.0; net.cos_embedding.weight[100, 7] = -80.0
        elif case == "one env huge among small":
            obs *= 1e-3; obs[17] = 1e5
    n = obs.shape[0]
    bc = row.view(1, 32).expand(n, 32).contiguous()
    with torch.no_grad():
        ref = copy.deepcopy(net).double().get_qvals(obs.double(), 1.0, taus=bc.double())
    _, q0 = fused_act(net, obs, 0.0, 1.0, taus=bc, want_qvals=True)
    _, q1 = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)
    assert bool(torch.isfinite(q1).all())
    row_scale = ref.abs().max(dim=1).values.clamp_min(1e-30)
    e0 = float(((q0.double() - ref).abs().max(dim=1).values / row_scale).max())
    e1 = float(((q1.double() - ref).abs().max(dim=1).values / row_scale).max())
    assert e1 < 2.0 * e0 + 1e-6, (case, e0, e1)


def test_act_eval_quantiles_and_taus(torch):
    """act_eval (agent.py:217-236): per-tau quantile values [n, 32, 9] and the taus they were evaluated at [n, 32, 1]."""
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    net = _net(torch, "pretrained")
    obs, row = _inputs(torch, 1000, 5.0)
    n = obs.shape[0]
    bc = row.view(1, 32).expand(n, 32).contiguous()
    with torch.no_grad():
        ref, _ = copy.deepcopy(net).double().forward(obs.double(), 32, 0.75, taus=bc.double())
    a0, z0, t0, q0 = fused_act(net, obs, 0.0, 0.75, taus=bc, want_quantiles=True, want_qvals=True)
    a1, z1, t1, q1 = fused_act(net, obs, 0.0, 0.75, taus=row, want_quantiles=True, want_qvals=True, shared_taus=True)
    assert z1.shape == (n, 32, 9) and t1.shape == (n, 32, 1)
    assert torch.equal(t1, t0)
    # random observations drive this network to |Z| ~ 1e3: float32-class = error relative to the largest quantile value, against float64,
    # no worse than the per-row kernel's
    scale = float(ref.abs().max())
    e0, e1 = float((z0.double() - ref).abs().max()) / scale, float((z1.double() - ref).abs().max()) / scale
    assert e1 < 3e-6 and e1 < 1.5 * e0 + 1e-7, (e0, e1, scale)
    assert float((q1.double() - z1.double().mean(dim=1)).abs().max()) / scale < 1e-6
    assert bool((a1.long() == q1.argmax(1)).all())


def test_results_do_not_depend_on_batch_size_or_position_and_modes_do_not_disturb_each_other(torch):
    from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
    net = _net(torch, "seeded")
    obs, row = _inputs(torch, 5000, 5.0)
    bc = row.view(1, 32).expand(5000, 32).contiguous()
    _, q_per0 = fused_act(net, obs, 0.0, 1.0, taus=bc, want_qvals=True)
    _, q_all = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)
    for lo, hi in ((0, 1), (17, 18), (100, 613), (4000, 5000)):
        _, q = fused_act(net, obs[lo:hi].contiguous(), 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)
        assert torch.equal(q, q_all[lo:hi])
    _, q_per1 = fused_act(net, obs, 0.0, 1.0, taus=bc, want_qvals=True)      # back in the default mode: the same bits as before
    assert torch.equal(q_per0, q_per1) and act_context(net).tau_mode == 0
    # a weight change is picked up (layer-1 constant AND weight image)
    with torch.no_grad():
        net.cos_embedding.weight.mul_(1.5); net.hidden_layer_2.bias.add_(0.25)
        ref = net.get_qvals(obs, 1.0, taus=bc)
    _, q2 = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)
    np.testing.assert_allclose(q2.cpu().numpy(), ref.cpu().numpy(), rtol=1e-5, atol=1e-5)
    assert not torch.equal(q2, q_all)


def test_library_draw_path(torch):
    """`mn_iqn_act_rng` in shared mode: draws[0 .. 32) = the launch's taus = U[0, 1) cvar, draws[32 .. 32 + n) = exploration uniforms;
    the call counter advances; actions are the epsilon-greedy choice on the Q-values of exactly those taus."""
    from distributional_rl_navigation_amd.iqn.fused_act import ActRng, fused_act
    net = _net(torch, "pretrained")
    obs, _ = _inputs(torch, 20000, 5.0)
    n = obs.shape[0]
    rng = ActRng(123, DEV)
    a, q = fused_act(net, obs, 0.0, 0.5, rng=rng, want_qvals=True, shared_taus=True)
    assert int(rng.state[1]) == 1
    row = rng.draws(n, 32)[:32].clone()
    assert float(row.min()) >= 0.0 and float(row.max()) < 0.5 and float(row.max()) > 0.25
    # the same taus injected (cvar already applied) give the same Q-values bit for bit
    _, q_inj = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)
    assert torch.equal(q, q_inj) and bool((a.long() == q.argmax(1)).all())
    # next call: other taus; a fresh generator with the same seed reproduces the first call
    _, _, t2 = fused_act(net, obs, 0.0, 0.5, rng=rng, want_quantiles=True, shared_taus=True)
    assert int(rng.state[1]) == 2 and t2.shape == (n, 32, 1) and bool((t2 == t2[0:1]).all()) and not torch.equal(t2[0, :, 0], row)
    rng_b = ActRng(123, DEV)
    a_b, _, t_b = fused_act(net, obs, 0.0, 0.5, rng=rng_b, want_quantiles=True, shared_taus=True)      # (act_eval's kernel: the same draws)
    assert torch.equal(t_b[0, :, 0], row) and float((a_b != a).float().mean()) < 1e-3
    # exploration
    a_e = fused_act(net, obs, 1.0, 1.0, rng=rng, shared_taus=True)
    cnt = torch.bincount(a_e.long(), minlength=9).float() / n
    assert bool(((a_e >= 0) & (a_e < 9)).all()) and float((cnt - 1 / 9).abs().max()) < 0.01
    a_g = fused_act(net, obs, 0.3, 1.0, rng=rng, shared_taus=True)
    u = rng.draws(n, 32)[32:32 + n]
    assert 0.27 < float((u <= 0.3).float().mean()) < 0.33
    # per-row cvar (adaptive policies) keeps per-row taus even when shared taus are asked for
    cv = torch.rand(n, device=DEV) * 0.9 + 0.1
    _, _, t_ad = fused_act(net, obs, 0.0, cv, rng=rng, want_quantiles=True, shared_taus=True)
    assert bool((t_ad[:, :, 0] <= cv.view(-1, 1)).all()) and not bool((t_ad[0] == t_ad[1]).all())


def test_agent_switch_and_closed_loop(torch):
    """`IQNAgent.shared_taus = True`: act_batch / act_eval_batch / vec_step run the shared-tau kernel; greedy evaluation of the shipped
    model on the 30 evaluation worlds reaches the same outcome class as with per-row taus (the policy is robust to the tau draw)."""
    import json
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    res = {}
    for shared in (False, True):
        agent = IQNAgent(26, 9, device=DEV, seed=3)
        agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), DEV)
        agent.shared_taus = shared
        env = VecMarineNavEnv(len(cfg), seed=0, device=DEV, precision="f64")
        r = agent.evaluation_vec(env, cfg, greedy=True)
        res[shared] = (int(np.sum(r["successes"])), float(np.mean(r["rewards"])))
        env.close()
    assert abs(res[True][0] - res[False][0]) <= 2 and abs(res[True][1] - res[False][1]) < 10.0, res
    agent = IQNAgent(26, 9, device=DEV, seed=1, BATCH_SIZE=64, BUFFER_SIZE=4096, learning_starts=1)
    agent.shared_taus = True
    venv = VecMarineNavEnv(512, seed=0, device=DEV, precision="f64")
    agent.learn_vec(total_vector_steps=8, train_env=venv, verbose=False)
    assert agent.grad_steps > 0
    venv.close()


def test_argument_checks(torch):
    from distributional_rl_navigation_amd import _capi
    from distributional_rl_navigation_amd.iqn.fused_act import act_context, _p
    net = _net(torch, "seeded")
    ctx = act_context(net)
    lib = _capi.lib()
    assert lib.mn_iqn_set_tau_mode(ctx.h, 4) != 0 and lib.mn_iqn_set_tau_mode(None, 1) != 0
    obs = torch.zeros(8, 26, device=DEV); row = torch.rand(32, device=DEV); q = torch.empty(8, 9, device=DEV)
    stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
    try:
        ctx.set_tau_mode(1)
        ctx.set_variant(0)      # only the split-f16 kernel has the shared-tau form
        assert lib.mn_iqn_act(ctx.h, _p(obs), _p(row), ctx.weights(net), _p(q), None, C.c_float(0.0), None, None, 8, 32, stream) != 0
        ctx.set_variant(2)
        assert lib.mn_iqn_act(ctx.h, _p(obs), _p(row), ctx.weights(net), _p(q), None, C.c_float(0.0), None, None, 8, 32, stream) == 0
        st = torch.tensor([1, 0], dtype=torch.int64, device=DEV); draws = torch.empty(8 * 33, device=DEV); cv = torch.ones(8, device=DEV)
        act = torch.empty(8, dtype=torch.int32, device=DEV)
        assert lib.mn_iqn_act_rng(ctx.h, _p(obs), ctx.weights(net), _p(st), _p(draws), _p(cv), C.c_float(1.0), C.c_float(0.0), _p(act), None, None, 8, 32, stream) != 0
        assert lib.mn_iqn_act_rng(ctx.h, _p(obs), ctx.weights(net), _p(st), _p(draws), None, C.c_float(1.0), C.c_float(0.0), _p(act), None, None, 8, 32, stream) == 0
        torch.cuda.synchronize()
    finally:
        ctx.set_variant(ctx.DEFAULT_VARIANT)
        ctx.set_tau_mode(0)


# ---- the environment-tiled form (csrc/iqn_act_tiled.h): what mode 1 runs from 65 536 rows up; pinned here with shared_taus="tiled" (mode 3) ----------------------------------------------------
def _f64_ref(torch, net, obs, row):
    n = obs.shape[0]
    bc = row.view(1, 32).expand(n, 32).contiguous()
    with torch.no_grad():
        return copy.deepcopy(net).double().get_qvals(obs.double(), 1.0, taus=bc.double()), bc


@pytest.mark.parametrize("which", ["seeded", "pretrained"])
def test_env_tiled_kernel_is_float32_class_and_agrees_with_the_wavefront_per_row_form(torch, which):
    from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
    net = _net(torch, which)
    obs, row = _inputs(torch, 16384 + 37, 5.0)      # ragged: the last workgroup's waves are partly / entirely past the end
    ref, bc = _f64_ref(torch, net, obs, row)
    ctx = act_context(net)
    try:
        ctx.set_variant(0)
        _, qe = fused_act(net, obs, 0.0, 1.0, taus=bc, want_qvals=True)
    finally:
        ctx.set_variant(ctx.DEFAULT_VARIANT)
    aw, qw = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus="wave")
    at, qt = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus="tiled")
    assert ctx.tau_mode == 3 and not torch.equal(qw, qt)      # (another kernel ran: another rounding)

    def err(q):
        d = (q.double() - ref).abs()
        return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())
    (mx_e, rms_e), (mx_t, rms_t) = err(qe), err(qt)
    assert rms_t < 1.25 * rms_e + 2e-8 and mx_t < 1.5 * mx_e + 1e-7, (mx_e, rms_e, mx_t, rms_t)
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5 * float(ref.abs().max())
    assert torch.equal(at.long()[clear], ref.argmax(dim=1)[clear]) and bool((at.long() == qt.argmax(1)).all())
    # rows do not depend on the batch they are in
    _, q_more = fused_act(net, torch.cat([obs, obs[:5000]]).contiguous(), 0.0, 1.0, taus=row, want_qvals=True, shared_taus="tiled")
    assert torch.equal(q_more[:obs.shape[0]], qt) and torch.equal(q_more[obs.shape[0]:], qt[:5000])


@pytest.mark.parametrize("case", ["obs x 1e6", "obs x 1e-6", "obs zero", "weights x 30", "weights x 1e-3", "one huge weight", "one env huge among small",
                                  "tiny layer-1 bounds"])
def test_env_tiled_kernel_range_scaling_cases(torch, case):
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    net = _net(torch, "seeded")
    obs, row = _inputs(torch, 16384, 5.0)
    with torch.no_grad():
        if case == "obs x 1e6": obs *= 1e6
        elif case == "obs x 1e-6": obs *= 1e-6
        elif case == "obs zero": obs.zero_()
        elif case == "weights x 30":
            for p in net.parameters():
                if p.dim() == 2: p.mul_(30.0)
        elif case == "weights x 1e-3":
            for p in net.parameters():
                if p.dim() == 2: p.mul_(1e-3)
        elif case == "one huge weight":
            net.hidden_layer.weight[3, 100] = 500.0; net.cos_embedding.weight[100, 7] = -80.0
        elif case == "one env huge among small":
            obs *= 1e-3; obs[17] = 1e5
        elif case == "tiny layer-1 bounds":      # B1_j << 1: the feature operand must be scaled by max |f|, not by the activation bound
            net.cos_embedding.weight.mul_(1e-4); net.cos_embedding.bias.mul_(1e-4); obs *= 100.0
    ref, bc = _f64_ref(torch, net, obs, row)
    _, q0 = fused_act(net, obs, 0.0, 1.0, taus=bc, want_qvals=True)
    _, q1 = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus="tiled")
    assert bool(torch.isfinite(q1).all())
    row_scale = ref.abs().max(dim=1).values.clamp_min(1e-30)
    e0 = float(((q0.double() - ref).abs().max(dim=1).values / row_scale).max())
    e1 = float(((q1.double() - ref).abs().max(dim=1).values / row_scale).max())
    assert e1 < 2.0 * e0 + 1e-6, (case, e0, e1)


def test_env_tiled_kernel_library_draws_and_exploration(torch):
    from distributional_rl_navigation_amd.iqn.fused_act import ActRng, fused_act
    net = _net(torch, "pretrained")
    obs, _ = _inputs(torch, 30000, 5.0)
    n = obs.shape[0]
    rng = ActRng(77, DEV)
    a, q = fused_act(net, obs, 0.0, 0.5, rng=rng, want_qvals=True, shared_taus="tiled")
    assert int(rng.state[1]) == 1
    row = rng.draws(n, 32)[:32].clone()
    _, q_inj = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus="tiled")
    assert torch.equal(q, q_inj) and bool((a.long() == q.argmax(1)).all())
    a_e = fused_act(net, obs, 1.0, 1.0, rng=rng, shared_taus="tiled")
    cnt = torch.bincount(a_e.long(), minlength=9).float() / n
    assert bool(((a_e >= 0) & (a_e < 9)).all()) and float((cnt - 1 / 9).abs().max()) < 0.01
    a_g = fused_act(net, obs, 0.3, 1.0, rng=rng, shared_taus="tiled")
    u = rng.draws(n, 32)[32:32 + n]
    a_greedy = fused_act(net, obs, 0.0, 1.0, taus=rng.draws(n, 32)[:32].clone(), shared_taus="tiled")
    keep = u > 0.3
    assert torch.equal(a_g[keep], a_greedy[keep]) and 0.67 < float(keep.float().mean()) < 0.73


def test_mode_1_switches_to_the_env_tiled_form_at_65536_rows(torch):
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    net = _net(torch, "pretrained")
    obs, row = _inputs(torch, 65536, 5.0)
    _, q_auto = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)
    _, q_tiled = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus="tiled")
    _, q_wave = fused_act(net, obs, 0.0, 1.0, taus=row, want_qvals=True, shared_taus="wave")
    assert torch.equal(q_auto, q_tiled) and not torch.equal(q_auto, q_wave)
    _, q_small = fused_act(net, obs[:30000].contiguous(), 0.0, 1.0, taus=row, want_qvals=True, shared_taus=True)
    assert torch.equal(q_small, q_wave[:30000])
    scale = float(q_wave.abs().max())
    assert float((q_tiled - q_wave).abs().max()) / scale < 3e-6
