"""CPU restatement (numpy) of the arithmetic the default act kernel relies on (csrc/iqn_act_split.h): the two-piece f16 split,
the three-product accumulation and the power-of-two range scaling chosen from a guaranteed bound.  No GPU: these tests pin the
SCHEME (error class, no-overflow guarantee) on the shipped network and on adversarial inputs; the kernel itself is checked against
a float64 evaluation in tests/test_act_split_gpu.py."""
import os

import numpy as np
import pytest

G = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def split(x):
    """hi = RNE16(x), lo = RNE16(x - hi) (x float32, |x| < 65504)."""
    x = np.asarray(x, dtype=np.float32)
    hi = x.astype(np.float16)
    lo = (x - hi.astype(np.float32)).astype(np.float16)        # the subtraction is exact in float32
    return hi, lo


def test_two_f16_pieces_carry_a_float32():
    rng = np.random.RandomState(0)
    x = (rng.standard_normal(200000) * np.exp2(rng.uniform(-1, 15, 200000))).astype(np.float32)
    x = x[np.abs(x) < 60000.0]
    hi, lo = split(x)
    rec = hi.astype(np.float64) + lo.astype(np.float64)
    err = np.abs(rec - x.astype(np.float64))
    # worst case 2^-22 relative (half an ulp of hi's 11 bits, then half an ulp of lo's 11 bits), 2^-25 absolute below that;
    # the rms is what matters for a dot product: ~2^-24.4 relative
    assert np.all(err <= np.maximum(np.abs(x) * 2.0 ** -22, 2.0 ** -25))
    big = np.abs(x) >= 1.0
    assert np.sqrt(np.mean((err[big] / np.abs(x[big])) ** 2)) < 2.0 ** -24
    # the residual x - hi is exactly representable in float32 (Sterbenz-type): recomputing it in float64 changes nothing
    assert np.array_equal((x - hi.astype(np.float32)).astype(np.float64), x.astype(np.float64) - hi.astype(np.float64))


def test_three_products_have_the_error_of_one_float32_rounding_per_product():
    rng = np.random.RandomState(1)
    K = 224
    A = (0.1 * rng.standard_normal((16, K))).astype(np.float32)
    B = np.maximum(rng.standard_normal((K, 16)), 0).astype(np.float32) * 37.0
    sa = np.float32(2.0 ** np.floor(np.log2(32768.0 / np.abs(A).max())))
    sb = np.float32(2.0 ** np.floor(np.log2(32768.0 / np.abs(B).max())))
    ah, al = split(A * sa)
    bh, bl = split(B * sb)
    f = lambda m: m.astype(np.float64)
    C3 = (f(al) @ f(bh) + f(ah) @ f(bl) + f(ah) @ f(bh)) / (float(sa) * float(sb))     # f16 x f16 products are exact; sum in float64
    C = f(A) @ f(B)
    bound = 3 * 2.0 ** -22 * (np.abs(f(A)) @ np.abs(f(B)))          # rigorous worst case
    assert np.all(np.abs(C3 - C) <= bound + 1e-30)
    # the same product in straight float32 accumulation is no better
    C32 = np.zeros((16, 16), dtype=np.float32)
    for k in range(K):
        C32 += np.outer(A[:, k], B[k, :]).astype(np.float32)
    assert np.sqrt(np.mean((C3 - C) ** 2)) <= 1.5 * np.sqrt(np.mean((f(C32) - C) ** 2))


def _load_weights(which):
    import torch
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    net = ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), "cpu") if which == "pretrained" else ObsEncoder(26, 9, seed=5, device="cpu")
    sd = {k: v.detach().double().numpy() for k, v in net.state_dict().items()}
    return net, sd


@pytest.mark.parametrize("which,wscale,oscale", [("pretrained", 1.0, 5.0), ("seeded", 1.0, 5.0), ("seeded", 30.0, 5.0), ("pretrained", 1.0, 1e6),
                                                 ("seeded", 1e-3, 1e-6)])
def test_scale_from_the_guaranteed_bound_never_overflows_f16(which, wscale, oscale):
    """S_l = 2^(14 - floor(log2 M_l)) with M1 = m1 = max_j B1_j |f_j|, M2 = R2 m1 + beta2, M3 = R3 M2 + beta3: then S_l |h_l| < 2^15 for
    every hidden activation of layer l (evaluated here in float64), whatever the weights and the observation -- and the bounds are
    tight enough that every layer keeps its largest scaled activation far above the f16 precision floor."""
    net, sd = _load_weights(which)
    W1, b1 = sd["cos_embedding.weight"] * wscale, sd["cos_embedding.bias"]
    W2, b2 = sd["hidden_layer.weight"] * wscale, sd["hidden_layer.bias"]
    W3, b3 = sd["hidden_layer_2.weight"] * wscale, sd["hidden_layer_2.bias"]
    rng = np.random.RandomState(3)
    n = 512
    obs = rng.standard_normal((n, 26)) * oscale
    obs[:, 4:][rng.uniform(size=(n, 22)) < 0.4] = 0.0
    feats = np.concatenate([obs[:, :2] @ (sd["velocity_encoder.weight"] * wscale).T + sd["velocity_encoder.bias"],
                            obs[:, 2:4] @ (sd["goal_encoder.weight"] * wscale).T + sd["goal_encoder.bias"],
                            obs[:, 4:] @ (sd["sensor_encoder.weight"] * wscale).T + sd["sensor_encoder.bias"]], axis=1)       # [n, 208]
    B1 = np.abs(W1).sum(1) + np.abs(b1)
    R2, be2 = np.abs(W2).sum(1).max(), np.abs(b2).max()
    R3, be3 = np.abs(W3).sum(1).max(), np.abs(b3).max()
    m1 = (np.abs(feats) * B1).max(1)
    scale = lambda M: 2.0 ** (14 - np.floor(np.log2(np.clip(M, 1e-30, 1e30))))
    S1, S2, S3 = scale(m1), scale(R2 * m1 + be2), scale(R3 * (R2 * m1 + be2) + be3)
    taus = rng.uniform(size=(n, 32))
    cos = np.cos(taus[:, :, None] * np.pi * np.arange(64)[None, None, :])              # [n, 32, 64]
    h1 = np.maximum(cos @ W1.T + b1, 0) * feats[:, None, :]
    h2 = np.maximum(h1 @ W2.T + b2, 0)
    h3 = np.maximum(h2 @ W3.T + b3, 0)
    for h, S in ((h1, S1), (h2, S2), (h3, S3)):
        top = np.abs(h).max(axis=(1, 2)) * S
        assert np.all(top < 2.0 ** 15)
        # the largest scaled activation of a layer stays >= 2^-1 (17 binades below the f16 ceiling: no precision is lost to the
        # conservative bound) whenever the layer has any activation at all
        assert np.all((top >= 0.5) | (np.abs(h).max(axis=(1, 2)) == 0.0)), float(top.min())
