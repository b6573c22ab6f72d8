"""Attribute the time of the step / reset kernels to their parts with the ABLATION build of the library
(libmarinenav_hip_ablation.so, `make -C distributional_rl_navigation_amd/csrc ablation`: the only build that contains the
MN_SKIP / MN_RSKIP switches and mn_set_debug_skip).  The shipped library cannot skip work.

    python scripts/step_ablation.py [n_envs]
"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from distributional_rl_navigation_amd import _capi

ABL = os.path.join(os.path.dirname(_capi.LIB_PATH), "libmarinenav_hip_ablation.so")
L = C.CDLL(ABL)
assert L.mn_build_info() == 1
for name, res, args in _capi.SIGNATURES:
    fn = getattr(L, name); fn.restype, fn.argtypes = res, args
L.mn_set_debug_skip.argtypes = [C.c_void_p, C.c_int32]
_capi._lib = L            # the package's VecMarineNavEnv, driven by the ablation build (this script only)
from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv

n = int(sys.argv[1]) if len(sys.argv) > 1 else 65536
env = VecMarineNavEnv(n, seed=0, device="cuda:0", precision="mixed")
env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
env.reset()
g = torch.Generator(device="cuda:0"); g.manual_seed(0)
acts = [torch.randint(0, 9, (n,), device="cuda:0", dtype=torch.int32, generator=g) for _ in range(64)]
for t in range(300):      # steady state: episodes at all ages
    env.step(acts[t % 64]); env.reset_done()


snap = env.get_state()


def timed(fn, reps=60):
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    env.set_state(*snap)          # every measurement starts from the same steady-state snapshot
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        fn()
    ev[1].record(); torch.cuda.synchronize()
    return ev[0].elapsed_time(ev[1]) / reps * 1e3


print(f"{n} envs, last step finished {env.last_done_count()} episodes")
print("step kernel (mn_step back to back, us per launch):")
for mask, what in ((0, "full"), (1, "no sub-steps"), (2, "no sonar scan"), (8, "no obstacle rotation / work-list"), (16, "no beam stores"),
                   (1 | 2 | 8 | 16, "loads + pose/state stores only")):
    L.mn_set_debug_skip(env.h, mask)
    print(f"  {what:36s} {timed(lambda: env.step(acts[0])):7.2f}")
L.mn_set_debug_skip(env.h, 0)
env.set_state(*snap)
env.step(acts[1])
k = env.last_done_count()
print(f"reset kernel (mn_reset_done of the same {k} finished envs back to back, us per launch):")
for mask, what in ((0, "full"), (32, "no start/goal loop"), (64, "no core loop"), (128, "no obstacle loop"), (256, "no first-observation sonar"),
                   (512, "no RNG write-back"), (32 | 64 | 128 | 256 | 512, "RNG load + table / pose stores only")):
    L.mn_set_debug_skip(env.h, mask)
    L.mn_set_debug_skip(env.h, mask)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    torch.cuda.synchronize(); ev[0].record()
    for _ in range(100):
        env.reset_done()
    ev[1].record(); torch.cuda.synchronize()
    print(f"  {what:36s} {ev[0].elapsed_time(ev[1]) / 100 * 1e3:7.2f}")
