"""The classical baselines as HIP device code (csrc/mn_planners.h): `mn_planner_act` -- one policy step for a vector of observation
rows -- against the tensor formulation of planners.py (which tests/test_planners_cpu.py pins to the reference's APF.py / BA.py on golden
G9), and `mn_rollout_policy` -- whole episodes under the policy in one launch -- against the launch-per-step loop and through the
experiment sweep (run_experiments.py:100-190,213-282)."""
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


def _tensor_policy(kind):
    from distributional_rl_navigation_amd.planners import apf_act_batch, ba_act_batch
    return apf_act_batch if kind == "APF" else ba_act_batch


@pytest.mark.parametrize("kind", ["APF", "BA"])
def test_planner_kernel_on_g9_and_on_synthetic_observations(torch, kind):
    from distributional_rl_navigation_amd.planners import planner_act_batch
    Z = np.load(os.path.join(G, "g9_planners.npz"))
    a, w = Z["a"], Z["w"]
    obs32 = torch.from_numpy(Z["obs"]).float().to(DEV)      # what the step kernels hand a policy: float32 rows
    act = planner_act_batch(obs32, kind, a, w)
    ref = _tensor_policy(kind)(obs32.double(), a, w)
    assert act.dtype == torch.int32 and torch.equal(act.long(), ref)                      # bit-equal to planners.py on the same rows
    golden = torch.from_numpy(Z["apf" if kind == "APF" else "ba"]).to(DEV)
    assert float((act.long() != golden).float().mean()) < 0.01                              # (the reference saw the float64 rows: ties may flip)
    # synthetic rows incl. the special cases: no return, one / two / many returns, standing still, goal behind
    g = torch.Generator(device=DEV); g.manual_seed(4)
    n = 200_000
    obs = torch.randn(n, 26, device=DEV, generator=g) * 6.0
    keep = torch.rand(n, 11, device=DEV, generator=g) < torch.rand(n, 1, device=DEV, generator=g)
    obs[:, 4:] = obs[:, 4:] * keep.repeat_interleave(2, dim=1)
    obs[: n // 50, :2] = 0.0
    obs[n // 50: n // 25, :2] *= 1e-4
    act = planner_act_batch(obs, kind, a, w)
    ref = _tensor_policy(kind)(obs.double(), a, w)
    # the sums of the tensor formulation are formed in another order than the device function's loops: a decision can differ where two
    # candidates tie to the last bit
    assert float((act.long() != ref).float().mean()) < 2e-4
    assert int(act.min()) >= 0 and int(act.max()) <= 8


@pytest.mark.parametrize("kind", ["APF", "BA"])
@pytest.mark.parametrize("precision", ["f64", "mixed"])
def test_rollout_policy_equals_the_launch_per_step_loop(torch, kind, precision):
    """`mn_rollout_policy`: 300 worlds, each env's episode under the device-side policy in ONE launch = the loop (mn_planner_act,
    mn_step) bit for bit -- actions, rewards, done / info codes until the env finishes, the terminal observation, pose and counters;
    afterwards the env idles (reward 0, done 1, terminal info, action -1)."""
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    from distributional_rl_navigation_amd.planners import planner_act_batch
    n, T = 300, 400
    envs = [VecMarineNavEnv(n, seed=21, device=DEV, precision=precision, obs64=precision == "f64") for _ in range(2)]
    for e in envs:
        e.set_attrs(num_cores=6, num_obs=8, min_start_goal_dis=30.0, N=5)
        e.reset()
    a_tab, w_tab = envs[0].params.a[:], envs[0].params.w[:]
    tr = envs[0].rollout_policy(T, kind, trace=("obs", "reward", "done", "info", "action"))
    env = envs[1]
    alive = torch.ones(n, dtype=torch.bool, device=DEV)
    obs = env.obs.clone()
    final_obs = obs.clone()
    state_at_end = [None] * n
    for t in range(T):
        act = planner_act_batch(obs, kind, a_tab, w_tab)
        nobs, rew, done, info = env.step(act)
        assert torch.equal(tr["action"][t][alive], act[alive]) and bool((tr["action"][t][~alive] == -1).all())
        assert torch.equal(tr["reward"][t][alive], rew[alive]) and bool((tr["reward"][t][~alive] == 0).all())
        assert torch.equal(tr["done"][t][alive], done[alive]) and bool((tr["done"][t][~alive] == 1).all())
        assert torch.equal(tr["info"][t][alive], info[alive])
        assert torch.equal(tr["obs"][t][alive], nobs[alive])
        just = alive & done.bool()
        if bool(just.any()):
            final_obs[just] = nobs[just]
            s, ep, tot = env.get_state()
            for i in torch.nonzero(just).view(-1).tolist():
                state_at_end[i] = (s[i].copy(), int(ep[i]), int(tot[i]))
        alive = alive & ~done.bool()
        obs = nobs.clone()
        if not bool(alive.any()):
            assert bool((tr["done"][t + 1:] == 1).all()) and bool((tr["action"][t + 1:] == -1).all())
            break
    finished = ~alive
    assert int(finished.sum()) > n // 2
    assert torch.equal(tr["final_obs"][finished], final_obs[finished])
    if precision == "f64":      # the float64 copies (mn_enable_obs64) of a finished env are its TERMINAL observation, not a later idle step's
        f = finished.cpu().numpy()
        assert np.array_equal(envs[0].get_obs64()[f].astype(np.float32), final_obs[finished].cpu().numpy())
    s, ep, tot = envs[0].get_state()
    for i in torch.nonzero(finished).view(-1).tolist()[:100]:
        assert np.array_equal(s[i], state_at_end[i][0]) and int(ep[i]) == state_at_end[i][1] and int(tot[i]) == state_at_end[i][2]
    for e in envs:
        e.close()


def test_experiment_sweep_records_are_the_same_with_and_without_the_rollout(torch):
    from distributional_rl_navigation_amd.experiments import run_experiment
    a, _ = run_experiment(None, n_obs=8, n_cores=6, num=64, seed=15, policies=("APF", "BA"), classical_rollout=True)
    b, _ = run_experiment(None, n_obs=8, n_cores=6, num=64, seed=15, policies=("APF", "BA"), classical_rollout=False)
    for name in ("APF", "BA"):
        for key in ("success", "out_of_area", "time", "energy", "reward", "actions"):
            assert a[name][key] == b[name][key], (name, key)
        assert len(a[name]["computation_times"]) == len(b[name]["computation_times"]) == sum(len(x) for x in a[name]["actions"])
    assert 0 < sum(a["APF"]["success"]) + sum(a["BA"]["success"])


def test_argument_checks(torch):
    import ctypes as C
    from distributional_rl_navigation_amd import _capi
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    L = _capi.lib()
    obs = torch.zeros(4, 26, device=DEV); act = torch.zeros(4, dtype=torch.int32, device=DEV)
    t3 = (C.c_double * 3)(-0.4, 0.0, 0.4)
    p = lambda t: C.c_void_p(t.data_ptr())
    assert L.mn_planner_act(p(obs), 4, 3, t3, t3, p(act), None) != 0 and L.mn_planner_act(p(obs), 0, 1, t3, t3, p(act), None) != 0
    assert L.mn_planner_act(None, 4, 1, t3, t3, p(act), None) != 0 and L.mn_planner_act(p(obs), 4, 2, t3, t3, p(act), None) == 0
    env = VecMarineNavEnv(8, device=DEV)
    env.reset()
    assert L.mn_rollout_policy(env.h, 0, 1, p(env.obs), None, None, None, None, None, None) != 0
    assert L.mn_rollout_policy(env.h, 5, 0, p(env.obs), None, None, None, None, None, None) != 0
    assert L.mn_rollout_policy(env.h, 5, 2, p(env.obs), None, None, None, None, None, None) == 0
    torch.cuda.synchronize()
    env.close()
