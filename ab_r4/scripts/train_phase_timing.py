"""Where the time of one fused gradient step goes: phase stamps (100 MHz wall clock) of ONE target and ONE local workgroup
of iqn_train_fwdbwd, from a profiling build of csrc/iqn_train.hip (-DMN_TRAIN_PHASES; compiled here into /tmp, never
loaded by the package).  usage: python scripts/train_phase_timing.py [batch] [reps]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
B = int(sys.argv[1]) if len(sys.argv) > 1 else 256
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
so = "/tmp/libtrainph.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include",
                       "-DMN_TRAIN_PHASES", "-shared", f"{ROOT}/distributional_rl_navigation_amd/csrc/iqn_train.hip", "-o", so])
L = C.CDLL(so)
L.mn_iqn_train_workspace_floats.restype = C.c_int64
dev = "cuda:0"
FLAGS = int(os.environ.get("MN_TRAIN_FLAGS", "3"))      # 3 = stage the next batch + start from the staged one, 0 = draw and gather in the launch
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ring = (torch.randn(n, 26, device=dev, generator=g), torch.randn(n, 26, device=dev, generator=g),
        torch.randint(0, 9, (n, 1), device=dev, generator=g), torch.randn(n, 1, device=dev, generator=g),
        (torch.rand(n, 1, device=dev, generator=g) < 0.05).float())
P = 35785
local = torch.randn(P, device=dev, generator=g) * 0.05
target = local + 0.01 * torch.randn(P, device=dev, generator=g)
ws = torch.zeros(L.mn_iqn_train_workspace_floats(B), device=dev)
grad = torch.zeros(P, device=dev); m = torch.zeros(P, device=dev); v = torch.zeros(P, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev); loss = torch.zeros(1, device=dev)
rng = torch.tensor([12345, 0], dtype=torch.int64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
names = {0: "start", 0.1: "all requests issued", 0.2: "generator state / staging tag read", 0.3: "wave 0: transitions in LDS", 1: "all waves: transitions in LDS (barrier)", 2: "encoders + cos", 3: "layer 1", 4: "layer 2", 5: "layer 3", 6: "output layer",
         7: "granules published", 8: "TD targets in LDS (hand-off wait)", 9: "loss + dh3", 10: "dh2, dW3, dW4", 11: "dx, dW2", 12: "Hadamard",
         13: "dW1, encoder grads issued"}
acc = np.zeros((2, 32)); cnt = 0
for it in range(reps + 20):
    rc = L.mn_iqn_train_grad_sampled(p(ring[0]), p(ring[1]), p(ring[2]), p(ring[3]), p(ring[4]), C.c_int64(n), p(rng), None, None,
                                     p(local), p(target), p(ws), p(grad), p(loss), B, 8, C.c_float(0.99), FLAGS, None)
    assert rc == 0
    rc = L.mn_iqn_train_adam(p(local), p(grad), p(m), p(v), p(step), p(ws), B, C.c_double(1e-4), C.c_double(0.9), C.c_double(0.999),
                             C.c_double(1e-8), C.c_double(0.5), C.c_float(1.0), 0, None)
    assert rc == 0
    if it >= 20:
        torch.cuda.synchronize()
        out = (C.c_ulonglong * 64)()
        assert L.mn_iqn_train_debug_phases(out) == 0
        a = np.array(out[:], dtype=np.float64).reshape(2, 32)
        t0 = min(a[0, 0], a[1, 0])
        rel = np.where(a > 0, (a - t0) * 0.01, np.nan)      # us
        acc += np.nan_to_num(rel); cnt += 1
acc /= cnt
print(f"batch {B}: mean over {cnt} steps, microseconds since the earlier of the two workgroups' start (loss {float(loss):.4f})")
for role, label in ((0, "TARGET workgroup 0"), (1, f"LOCAL workgroup {B // 2}")):
    print(label)
    prev = None
    for k in sorted(names):
        if (role == 0 and k in (8, 9, 10, 11, 12, 13)) or (role == 1 and k == 7):
            continue
        t = acc[role, {0.1: 14, 0.2: 15, 0.3: 16}.get(k, k)]
        print(f"  {k:4.1f} {names[k]:42s} at {t:7.2f} us" + (f"   (+{t - prev:5.2f})" if prev is not None else ""))
        prev = t

out2 = (C.c_ulonglong * 32)()
assert L.mn_iqn_train_debug_phases2(out2) == 0
a2 = np.array(out2[:], dtype=np.float64).reshape(2, 2, 8)
for kern, label, nst in ((0, "iqn_grad_reduce", 4), (1, "iqn_adam", 3)):
    for blk, bl in ((0, "block 0"), (1, "middle block")):
        t = a2[kern, blk, :nst]
        print(f"{label} {bl}: " + "  ".join(f"+{(t[k] - t[0]) * 0.01:5.2f}" for k in range(nst)) + "  us since the block's start (last launch)")
print("reduce stamps: start | partial sums formed | after barrier | end;  adam stamps: start | norm + bias corrections | end")
