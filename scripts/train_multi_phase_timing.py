"""Where a step of the persistent multi-step launch (mn_iqn_train_steps) goes: 100 MHz stamps of target workgroup 0 and the first local workgroup for the last four
steps of a G-step launch, from a profiling build (-DMN_TRAIN_PHASES, compiled into /tmp).  usage: python scripts/train_multi_phase_timing.py [G] [reps]"""
import ctypes as C, os, subprocess, sys
import numpy as np, torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
G = int(sys.argv[1]) if len(sys.argv) > 1 else 16
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 100
B = 256
so = "/tmp/libtrainph.so"
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include", "-ffp-contract=off", "-mllvm", "-disable-machine-licm",
                       "-DMN_TRAIN_PHASES", *os.environ.get("MN_EXTRA_DEFS", "").split(), "-shared", f"{ROOT}/distributional_rl_navigation_amd/csrc/iqn_train.hip", "-o", so])
L = C.CDLL(so)
L.mn_iqn_train_workspace_floats.restype = C.c_int64
dev = "cuda:0"
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ring = (torch.randn(n, 26, device=dev, generator=g), torch.randn(n, 26, device=dev, generator=g), torch.randint(0, 9, (n, 1), device=dev, generator=g),
        torch.randn(n, 1, device=dev, generator=g), (torch.rand(n, 1, device=dev, generator=g) < 0.05).float())
P = 35785
local = torch.randn(P, device=dev, generator=g) * 0.05
target = local + 0.01 * torch.randn(P, device=dev, generator=g)
ws = torch.zeros(L.mn_iqn_train_workspace_floats(B), device=dev)
assert L.mn_iqn_train_workspace_init(C.c_void_p(ws.data_ptr()), B, None) == 0
grad = torch.zeros(P, device=dev); m = torch.zeros(P, device=dev); v = torch.zeros(P, device=dev)
step = torch.zeros(1, dtype=torch.int32, device=dev); losses = torch.zeros(max(G, 16), device=dev)
rng = torch.tensor([12345, 0], dtype=torch.int64, device=dev)
p = lambda t: C.c_void_p(t.data_ptr())
acc = np.zeros((2, 4, 16)); cnt = 0
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
tot = 0.0
for it in range(reps + 10):
    ev0.record()
    rc = L.mn_iqn_train_steps(p(ring[0]), p(ring[1]), p(ring[2]), p(ring[3]), p(ring[4]), C.c_int64(n), p(rng), None, None, p(local), p(target), p(ws), p(grad), p(losses),
                              p(m), p(v), p(step), B, 8, C.c_float(0.99), 3, G, C.c_double(1e-4), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-8), C.c_double(0.5), None)
    ev1.record()
    assert rc == 0
    torch.cuda.synchronize()
    if it >= 10:
        tot += ev0.elapsed_time(ev1)
        o = (C.c_ulonglong * 128)()
        assert L.mn_iqn_train_debug_stepph(o) == 0
        a = np.array(o[:], dtype=np.float64).reshape(2, 4, 16)
        # step indices of the last four steps of the launch: G - 4 .. G - 1, stored at slot (k & 3); reference = the local workgroup's "step begins" of step G - 3
        ks = [k for k in range(max(1, G - 3), G)]
        t0 = a[1, ks[0] & 3, 0]
        acc += np.where(a > 0, (a - t0) * 0.01, 0.0); cnt += 1
acc /= cnt
print(f"G = {G}: {1e3 * tot / reps / G:.2f} us per step (events, profiling build); stamps in us relative to the local workgroup's start of step {max(1, G - 3)} (loss {float(losses[G - 1]):.4f})")
for k in range(max(1, G - 3), G):
    lw, tw = acc[1, k & 3], acc[0, k & 3]
    print(f"  step {k}: local wg: begins {lw[0]:7.2f}  requests out {lw[1]:7.2f}  TD in LDS {lw[3]:7.2f}  last store issued {lw[4]:7.2f}  acknowledged {lw[5]:7.2f}  share done {lw[6]:7.2f}"
          f" | target wg 0: TD({k}) published {tw[2]:7.2f}  tail({k}) begins {tw[8]:7.2f}  parameters out {tw[12]:7.2f}")
