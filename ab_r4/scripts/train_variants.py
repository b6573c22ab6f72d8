"""A / B of compile-time variants of csrc/iqn_train.hip on one GPU (each variant compiled into /tmp here, driven through the
C-ABI directly): microseconds per gradient step, per [forward/backward + reduce] pair and per Adam launch, back to back.
usage: python scripts/train_variants.py "NAME:-DFLAG=.. -DFLAG2=.." ..."""
import ctypes as C
import os
import subprocess
import sys
import time

import torch

ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
variants = [v.split(":", 1) for v in sys.argv[1:]] or [["default", ""]]
B, reps = 256, 3000
dev = "cuda:0"
FLAGS = int(os.environ.get("MN_TRAIN_FLAGS", "3"))      # 3 = stage the next batch + start from the staged one, 0 = draw and gather in the launch
g = torch.Generator(device=dev); g.manual_seed(0)
n = 100_000
ring = (torch.randn(n, 26, device=dev, generator=g), torch.randn(n, 26, device=dev, generator=g),
        torch.randint(0, 9, (n, 1), device=dev, generator=g), torch.randn(n, 1, device=dev, generator=g),
        (torch.rand(n, 1, device=dev, generator=g) < 0.05).float())
P = 35785
p = lambda t: C.c_void_p(t.data_ptr())
for rnd in range(2):
    for name, flags in variants:
        so = f"/tmp/libtrain_{name}.so"
        if rnd == 0:
            subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", f"-I{ROOT}/include"] + flags.split() +
                                  ["-shared", f"{ROOT}/distributional_rl_navigation_amd/csrc/iqn_train.hip", "-o", so])
        L = C.CDLL(so)
        L.mn_iqn_train_workspace_floats.restype = C.c_int64
        g.manual_seed(1)
        local = torch.randn(P, device=dev, generator=g) * 0.05
        target = local + 0.01 * torch.randn(P, device=dev, generator=g)
        ws = torch.zeros(L.mn_iqn_train_workspace_floats(B), device=dev)
        grad = torch.zeros(P, device=dev); m = torch.zeros(P, device=dev); v = torch.zeros(P, device=dev)
        step = torch.zeros(1, dtype=torch.int32, device=dev); loss = torch.zeros(1, device=dev)
        rng = torch.tensor([12345, 0], dtype=torch.int64, device=dev)

        def grad_call():
            assert L.mn_iqn_train_grad_sampled(p(ring[0]), p(ring[1]), p(ring[2]), p(ring[3]), p(ring[4]), C.c_int64(n), p(rng), None, None,
                                               p(local), p(target), p(ws), p(grad), p(loss), B, 8, C.c_float(0.99), FLAGS, None) == 0

        def adam_call():
            assert L.mn_iqn_train_adam(p(local), p(grad), p(m), p(v), p(step), p(ws), B, C.c_double(1e-4), C.c_double(0.9), C.c_double(0.999),
                                       C.c_double(1e-8), C.c_double(0.5), C.c_float(1.0), 0, None) == 0

        def rate(fn):
            for _ in range(50):
                fn()
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(reps):
                fn()
            torch.cuda.synchronize()
            return 1e6 * (time.perf_counter() - t0) / reps
        full = rate(lambda: (grad_call(), adam_call()))
        gonly = rate(grad_call)
        aonly = rate(adam_call)
        print(f"round {rnd} {name:14s} step {full:6.2f} us ({1e6 / full:7.0f} /s)   grad+reduce {gonly:6.2f} us   adam {aonly:5.2f} us   loss {float(loss):.5f}", flush=True)
