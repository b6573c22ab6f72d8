"""Where the time of the one- / two-launch gradient step (mn_iqn_train_step) goes: 100 MHz wall-clock stamps of one target workgroup, one local
workgroup and the first / middle / last reduction + Adam block, from a profiling build of csrc/iqn_train.hip (-DMN_TRAIN_PHASES, compiled into
/tmp, never loaded by the package).  usage: python scripts/train_step_phase_timing.py [launches = 1 | 2] [reps]"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np
import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
launches = int(sys.argv[1]) if len(sys.argv) > 1 else 1
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 200
B = 256
so = "/tmp/libtrainph.so"
if not os.path.exists(so) or os.environ.get("REBUILD", "1") == "1":
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include",
                           "-DMN_TRAIN_PHASES", *os.environ.get("MN_EXTRA_DEFS", "").split(), "-shared", f"{ROOT}/distributional_rl_navigation_amd/csrc/iqn_train.hip", "-o", so])
L = C.CDLL(so)
L.mn_iqn_train_workspace_floats.restype = C.c_int64
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ring = (torch.randn(n, 26, device=dev, generator=g), torch.randn(n, 26, device=dev, generator=g),
        torch.randint(0, 9, (n, 1), device=dev, generator=g), torch.randn(n, 1, device=dev, generator=g),
        (torch.rand(n, 1, device=dev, generator=g) < 0.05).float())
P = 35785
local = torch.randn(P, device=dev, generator=g) * 0.05
target = local + 0.01 * torch.randn(P, device=dev, generator=g)
ws = torch.zeros(L.mn_iqn_train_workspace_floats(B), device=dev)
assert L.mn_iqn_train_workspace_init(C.c_void_p(ws.data_ptr()), B, None) == 0
grad = torch.zeros(P, device=dev); m = torch.zeros(P, device=dev); v = torch.zeros(P, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev); loss = torch.zeros(1, device=dev)
rng = torch.tensor([12345, 0], dtype=torch.int64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
flags = 3 | (4 if launches == 1 else 0) | int(os.environ.get("MN_STEP_FLAGS", "0"))      # (16 / 32 / 48: the XCD-misplacement test hooks)
acc1 = np.zeros((2, 32)); acc3 = np.zeros((3, 8)); accw = np.zeros((256, 4)); cnt = 0
for it in range(reps + 20):
    rc = L.mn_iqn_train_step(p(ring[0]), p(ring[1]), p(ring[2]), p(ring[3]), p(ring[4]), C.c_int64(n), p(rng), None, None, None, None, None,
                             p(local), p(target), p(ws), p(grad), p(loss), p(m), p(v), p(step), B, 8, C.c_float(0.99), flags,
                             C.c_double(1e-4), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-8), C.c_double(0.5), None)
    assert rc == 0
    if it >= 20:
        torch.cuda.synchronize()
        o1 = (C.c_ulonglong * 64)(); o3 = (C.c_ulonglong * 24)()
        assert L.mn_iqn_train_debug_phases(o1) == 0 and L.mn_iqn_train_debug_phases3(o3) == 0
        a1 = np.array(o1[:], dtype=np.float64).reshape(2, 32); a3 = np.array(o3[:], dtype=np.float64).reshape(3, 8)
        t0 = min(a1[0, 0], a1[1, 0])
        acc1 += np.where(a1 > 0, (a1 - t0) * 0.01, 0.0); acc3 += np.where(a3 > 0, (a3 - t0) * 0.01, 0.0); cnt += 1
        ow = (C.c_ulonglong * 4096)()
        assert L.mn_iqn_train_debug_wgt(ow) == 0
        accw += (np.array(ow[:1024], dtype=np.float64).reshape(256, 4) - t0) * 0.01
acc1 /= cnt; acc3 /= cnt; accw /= cnt
print(f"{launches} launch(es) per step, mean over {cnt} steps, microseconds since the first forward / backward workgroup's start (loss {float(loss):.4f})")
print(f"  target workgroup 0: start {acc1[0, 0]:6.2f}   TD targets published {acc1[0, 7]:6.2f}")
print(f"  local workgroup {B // 2}: start {acc1[1, 0]:6.2f}   TD targets in LDS {acc1[1, 8]:6.2f}   last gradient store issued {acc1[1, 13]:6.2f}"
      + (f"   stores acknowledged {acc1[1, 17]:6.2f}   group share done {acc1[1, 18]:6.2f}" if launches == 1 else ""))
for b, lab in enumerate(("first", "middle", "last")):
    t = acc3[b]
    print(f"  {lab:6s} reduction + Adam block: start {t[0]:6.2f}   rows complete {t[1]:6.2f}   column sums formed {t[2]:6.2f}   norm partials in {t[3]:6.2f}   Adam done {t[4]:6.2f}")
q = lambda x: "min %.2f  median %.2f  p90 %.2f  max %.2f" % (x.min(), np.median(x), np.percentile(x, 90), x.max())
print("  target workgroups: start " + q(accw[:128, 0]) + " | end " + q(accw[:128, 1]))
print("  local  workgroups: start " + q(accw[128:, 0]) + " | end " + q(accw[128:, 1]))
if launches == 1:
    print("  local  workgroups: row acknowledged " + q(accw[128:, 2]) + " | group's rows seen " + q(accw[128:, 3]))
late = np.argsort(accw[128:, 1])[-8:]
print("  the eight local workgroups that end last: " + ", ".join(f"{128 + i} ({accw[128 + i, 1]:.2f})" for i in late))
