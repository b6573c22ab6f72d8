"""The default acting kernel computes the float32 network on the f16 matrix pipe (every operand split into two f16 pieces,
three MFMA products per float32 product, power-of-two range scaling from a guaranteed bound --
distributional_rl_navigation_amd/csrc/iqn_act_split.h).  The claim under test: its results are float32-class, i.e. measured
against a FLOAT64 evaluation of the same network (thirdparty/IQN/model.py:160-191 via iqn/model.py) its error is the error
of float32 arithmetic -- no worse than the exact-f32 MFMA kernel's and eager PyTorch float32's -- for ordinary inputs and
for inputs chosen to stress the range scaling."""
import copy
import os

import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
SPLIT, EXACT = 2, 0


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("needs a GPU")
    return t


def _nets(torch, which):
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    dev = "cuda:0"
    net = ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), dev) if which == "pretrained" else ObsEncoder(26, 9, seed=5, device=dev)
    return net


def _errors(torch, net, obs, taus, variants=(EXACT, SPLIT)):
    from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
    ctx = act_context(net)
    net64 = copy.deepcopy(net).double()
    with torch.no_grad():
        ref = net64.get_qvals(obs.double(), 1.0, taus=taus.double())
        eager = net.get_qvals(obs, 1.0, taus=taus)
    out = {}
    try:
        for v in variants:
            ctx.set_variant(v)
            a, q = fused_act(net, obs, 0.0, 1.0, taus=taus, want_qvals=True)
            d = (q.double() - ref).abs()
            out[v] = dict(q=q, a=a, max=float(d.max() / ref.abs().max()), rms=float((d.pow(2).mean() / ref.pow(2).mean()).sqrt()))
    finally:
        ctx.set_variant(ctx.DEFAULT_VARIANT)
    d = (eager.double() - ref).abs()
    out["eager"] = dict(max=float(d.max() / ref.abs().max()), rms=float((d.pow(2).mean() / ref.pow(2).mean()).sqrt()))
    return ref, out


def _inputs(torch, n, scale, seed=11):
    g = torch.Generator(device="cuda:0"); g.manual_seed(seed)
    obs = torch.randn(n, 26, device="cuda:0", generator=g) * scale
    obs[:, 4:][torch.rand(n, 22, device="cuda:0", generator=g) < 0.4] = 0.0      # sonar misses are exact zeros
    return obs, torch.rand(n, 32, device="cuda:0", generator=g)


@pytest.mark.parametrize("which", ["seeded", "pretrained"])
def test_split_kernel_is_float32_class_against_float64(torch, which):
    net = _nets(torch, which)
    obs, taus = _inputs(torch, 16384, 5.0)
    ref, e = _errors(torch, net, obs, taus)
    # float32 arithmetic on this network lands at 1e-7 .. 7e-7 relative rms; the split kernel must sit with the exact one
    assert e[SPLIT]["rms"] < 1.25 * e[EXACT]["rms"] + 2e-8, e
    assert e[SPLIT]["max"] < 1.5 * e[EXACT]["max"] + 1e-7, e
    assert e[SPLIT]["rms"] < 1e-6 and e[SPLIT]["max"] < 3e-6, e
    assert e[SPLIT]["rms"] < 3.0 * e["eager"]["rms"], e
    top2 = ref.topk(2, dim=1).values
    clear = (top2[:, 0] - top2[:, 1]) > 1e-5 * float(ref.abs().max())
    assert torch.equal(e[SPLIT]["a"].long()[clear], ref.argmax(dim=1)[clear])


@pytest.mark.parametrize("case", ["obs x 1e3", "obs x 1e6", "obs x 1e-6", "obs zero", "weights x 30", "weights x 1e-3", "one huge weight",
                                  "one env huge among small"])
def test_range_scaling_is_safe_and_accurate(torch, case):
    """The per-environment power-of-two scale comes from a guaranteed bound on the activations, so no input can overflow the
    f16 pieces, and because the bound is conservative the test also checks that accuracy survives it."""
    net = _nets(torch, "seeded")
    obs, taus = _inputs(torch, 4096, 5.0)
    with torch.no_grad():
        if case == "obs x 1e3": obs *= 1e3
        elif case == "obs x 1e6": obs *= 1e6
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
    ref, e = _errors(torch, net, obs, taus)
    assert bool(torch.isfinite(e[SPLIT]["q"]).all())
    assert e[SPLIT]["rms"] < 1.5 * e[EXACT]["rms"] + 5e-8, (case, e)
    # per-row accuracy (the scale is per environment: a huge neighbour must not cost a small row its precision)
    row_scale = ref.abs().max(dim=1).values.clamp_min(1e-30)
    row_err = ((e[SPLIT]["q"].double() - ref).abs().max(dim=1).values / row_scale)
    row_err0 = ((e[EXACT]["q"].double() - ref).abs().max(dim=1).values / row_scale)
    assert float(row_err.max()) < 2.0 * float(row_err0.max()) + 1e-6, (case, float(row_err.max()), float(row_err0.max()))


def test_results_do_not_depend_on_batch_position_or_size(torch):
    """One wavefront per environment, nothing shared between environments: row i of a batch equals the same observation
    evaluated alone, bit for bit, and repeated calls are bit-identical."""
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    net = _nets(torch, "pretrained")
    obs, taus = _inputs(torch, 8192 + 5, 6.0, seed=3)
    _, q = fused_act(net, obs, 0.0, 1.0, taus=taus, want_qvals=True)
    _, q2 = fused_act(net, obs, 0.0, 1.0, taus=taus, want_qvals=True)
    assert torch.equal(q, q2)
    for lo, hi in ((0, 1), (4000, 4007), (8192, 8197)):
        _, qs = fused_act(net, obs[lo:hi].contiguous(), 0.0, 1.0, taus=taus[lo:hi].contiguous(), want_qvals=True)
        assert torch.equal(qs, q[lo:hi])


def test_weight_image_follows_weight_changes(torch):
    """The split image carries scale constants derived from the weights (max |W|, row-sum bounds): they are rebuilt with the
    image when the weights change, including a change of magnitude that moves every power-of-two scale."""
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act, weights_changed
    net = _nets(torch, "seeded")
    obs, taus = _inputs(torch, 2048, 5.0)
    _, q0 = fused_act(net, obs, 0.0, 1.0, taus=taus, want_qvals=True)
    with torch.no_grad():
        for p in net.parameters():
            p.mul_(7.3 if p.dim() == 2 else 0.5)
    weights_changed(net)
    ref, e = _errors(torch, net, obs, taus, variants=(SPLIT,))
    assert e[SPLIT]["rms"] < 1e-6 and not torch.allclose(e[SPLIT]["q"], q0)


def test_act_rng_path_and_exploration_on_the_split_kernel(torch):
    """The default act path (in-library counter-based taus + exploration draws, agent.py:186-205): greedy where u > eps, uniform
    random actions elsewhere, and the Q-values it reports are those of its own taus."""
    from distributional_rl_navigation_amd.iqn.fused_act import ActRng, fused_act
    net = _nets(torch, "pretrained")
    n = 20000
    obs, _ = _inputs(torch, n, 4.0, seed=5)
    rng = ActRng(123, "cuda:0")
    a, q = fused_act(net, obs, 0.3, 1.0, rng=rng, want_qvals=True)
    draws = rng.draws(n, net.K)
    taus, u = draws[:n * net.K].view(n, net.K).clone(), draws[n * net.K:].clone()
    with torch.no_grad():
        ref = net.get_qvals(obs, 1.0, taus=taus)
    assert float((q - ref).abs().max()) < 3e-5 * float(ref.abs().max())
    greedy = u > 0.3
    top2 = ref.topk(2, dim=1).values
    clear = greedy & ((top2[:, 0] - top2[:, 1]) > 1e-4 * float(ref.abs().max()))
    assert torch.equal(a.long()[clear], ref.argmax(dim=1)[clear])
    frac = float((a.long() == ref.argmax(dim=1)).float().mean())
    assert 0.70 < frac < 0.77 and bool(((a >= 0) & (a < 9)).all())
    assert int(rng.state[1]) == 1          # the act kernel advanced the call counter


def test_greedy_policy_is_the_same_on_the_split_and_the_exact_kernel(torch, tmp_path):
    """Policy-level equivalence: the shipped IQN model evaluated greedily on the reference's 30 evaluation worlds, once acting
    through the exact-f32 kernel and once through the split-f16 kernel, with the same tau draws (same generator seed and call
    sequence).  Q-values differ by float32 rounding only, so the episodes are the same action for action unless a decision is a
    tie at the 1e-7 level."""
    import json
    import numpy as np
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.iqn.fused_act import act_context
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    res = {}
    for variant in (EXACT, SPLIT):
        agent = IQNAgent(26, 9, device="cuda:0", seed=0, BUFFER_SIZE=1024)
        agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
        act_context(agent.qnetwork_local).set_variant(variant)
        env = VecMarineNavEnv(30, device="cuda:0", precision="f64")
        res[variant] = agent.evaluation_vec(env, cfg, greedy=True, eval_log_path=None)
        env.close()
    same = sum(a == b for a, b in zip(res[EXACT]["actions"], res[SPLIT]["actions"]))
    assert same >= 28, (same, res[EXACT]["successes"], res[SPLIT]["successes"])
    assert abs(sum(res[EXACT]["successes"]) - sum(res[SPLIT]["successes"])) <= 1
    if same == 30:
        assert np.allclose(res[EXACT]["rewards"], res[SPLIT]["rewards"], rtol=0, atol=1e-9)
    print(f"identical episodes: {same}/30; successes exact {sum(res[EXACT]['successes'])}, split {sum(res[SPLIT]['successes'])}")


def test_quantile_output_of_the_split_kernel(torch):
    """`act_eval` (agent.py:217-236) on the split kernel: the per-tau quantile values Z(tau, a) against a float64 evaluation of the
    network, against the exact kernel's, and Q as their mean over the 32 taus."""
    from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
    for which in ("seeded", "pretrained"):
        net = _nets(torch, which)
        net64 = copy.deepcopy(net).double()
        obs, taus = _inputs(torch, 4099, 5.0, seed=17)
        with torch.no_grad():
            ref, _ = net64.forward(obs.double(), net.K, 1.0, taus=taus.double())
        ctx = act_context(net)
        out = {}
        try:
            for v in (EXACT, SPLIT):
                ctx.set_variant(v)
                out[v] = fused_act(net, obs, 0.0, 1.0, taus=taus, want_quantiles=True, want_qvals=True)
        finally:
            ctx.set_variant(ctx.DEFAULT_VARIANT)
        a0, z0, t0, q0 = out[EXACT]
        a2, z2, t2, q2 = out[SPLIT]
        scale = float(z0.abs().max())
        assert z2.shape == (4099, 32, 9) and bool(torch.isfinite(z2).all())
        assert float((z2 - z0).abs().max()) < 3e-6 * scale
        assert float((z2.mean(dim=1) - q2).abs().max()) < 2e-6 * scale and float((q2 - q0).abs().max()) < 3e-6 * scale
        assert torch.equal(t2, taus.view(4099, 32, 1))
        if True:
            e0 = float(((z0.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
            e2 = float(((z2.double() - ref).pow(2).mean() / ref.pow(2).mean()).sqrt())
            assert e2 < 1.25 * e0 + 2e-8, (which, e0, e2)
