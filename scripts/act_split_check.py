"""Accuracy and speed of the acting-kernel variants against a float64 evaluation of the same network (MI355X).
variant 0 = exact-f32 MFMA 16x16x4, 2 = split-f16 (three f16 MFMA products per f32 product)."""
import copy, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd.iqn.fused_act import act_context, fused_act
from distributional_rl_navigation_amd.iqn.model import ObsEncoder

G = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
dev = "cuda:0"
variants = [int(v) for v in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,2".split(","))]


def rel(q, ref):
    d = (q.double() - ref).abs()
    return float(d.max() / ref.abs().max()), float((d.pow(2).mean() / ref.pow(2).mean()).sqrt())


for name in ("seeded", "pretrained", "seeded x obs 1e3", "seeded x weights 30"):
    net = ObsEncoder.load(os.path.join(G, "pretrained_IQN_seed3"), dev) if name == "pretrained" else ObsEncoder(26, 9, seed=5, device=dev)
    if name.endswith("weights 30"):
        with torch.no_grad():
            for p in net.parameters():
                p.mul_(30.0 if p.dim() == 2 else 1.0)
    ctx = act_context(net)
    net64 = copy.deepcopy(net).double()
    g = torch.Generator(device=dev); g.manual_seed(11)
    n = 16384
    obs = torch.randn(n, 26, device=dev, generator=g) * (5e3 if "obs 1e3" in name else 5.0)
    obs[:, 4:][torch.rand(n, 22, device=dev, generator=g) < 0.4] = 0.0
    taus = torch.rand(n, 32, device=dev, generator=g)
    with torch.no_grad():
        ref = net64.get_qvals(obs.double(), 1.0, taus=taus.double())
        eager = net.get_qvals(obs, 1.0, taus=taus)
    print(f"{name}: max |Q| = {float(ref.abs().max()):.3f}")
    print("   eager torch f32      max err / max|Q| = %.3e   rms err / rms Q = %.3e" % rel(eager, ref))
    for v in variants:
        ctx.set_variant(v)
        a, q = fused_act(net, obs, 0.0, 1.0, taus=taus, want_qvals=True)
        e = rel(q, ref)
        agree = float((a.long() == ref.argmax(dim=1)).float().mean())
        print(f"   variant {v}            max err / max|Q| = {e[0]:.3e}   rms err / rms Q = {e[1]:.3e}   argmax agreement with f64 {agree:.6f}   finite {bool(torch.isfinite(q).all())}")
    ctx.set_variant(0)

# timing at the headline batch
net = ObsEncoder(26, 9, seed=5, device=dev); ctx = act_context(net)
n = 65536
obs = torch.randn(n, 26, device=dev) * 5.0; taus = torch.rand(n, 32, device=dev)
for v in variants:
    ctx.set_variant(v)
    for _ in range(5): fused_act(net, obs, 0.0, 1.0, taus=taus)
    ctx.profile_begin(50)
    for _ in range(50): fused_act(net, obs, 0.0, 1.0, taus=taus)
    ms, k = ctx.profile_end()
    print(f"variant {v}: {ms * 1e3:.1f} us per act call at {n} envs ({k} launches)")
