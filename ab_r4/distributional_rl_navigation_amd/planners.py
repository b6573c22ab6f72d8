"""Classical baselines on device: batched artificial-potential-field and bug-algorithm policies.

Counterparts of the reference's APF.py (:17-78) and BA.py (:14-155), which map ONE 26-dim observation
to an action index with python loops; here a whole vector of observations [n, 26] (device tensor) is
mapped at once with tensor ops, so the comparison table of run_experiments.py can be produced on the
GPU next to the IQN policies.  Both are stateless obs -> action maps.
"""
import math

import torch


def planner_act_batch(obs, kind, a, w):
    """One policy step of the classical baselines for obs [N, 26] float32 ON THE GPU as one HIP launch (C-ABI mn_planner_act: the device
    functions of csrc/mn_planners.h, float64 arithmetic on the float32 observation rows) -> actions [N] int32.  `kind`: "APF" | "BA".
    The tensor formulations below (`apf_act_batch`, `ba_act_batch`) are the same maps in PyTorch ops: the definition the kernel is tested
    against, and what runs on CPU tensors."""
    import ctypes as C
    from . import _capi
    assert obs.is_cuda and obs.dtype == torch.float32
    obs = obs.contiguous()
    n = obs.shape[0]
    out = torch.empty(n, dtype=torch.int32, device=obs.device)
    if n == 0:
        return out
    at = (C.c_double * 3)(*[float(v) for v in a])
    wt = (C.c_double * 3)(*[float(v) for v in w])
    rc = _capi.lib().mn_planner_act(C.c_void_p(obs.data_ptr()), n, {"APF": 1, "BA": 2}[kind], at, wt, C.c_void_p(out.data_ptr()),
                                    C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
    if rc:
        raise _capi.MarineNavHipError(f"mn_planner_act failed ({rc})")
    return out


def _wrap_to_pi(a):
    """BA.py:157-162 / APF.py:55-59: wrap to [-pi, pi).  Arguments are differences of two atan2 values
    (or one +- a margin), i.e. inside (-3pi, 3pi): two conditional shifts reproduce the reference's while
    loops exactly, and values already in range pass through bit-unchanged (a remainder would round away the
    1e-17-sized differences that decide `diff_angle > 0`)."""
    for _ in range(2):
        a = torch.where(a < -math.pi, a + 2 * math.pi, torch.where(a >= math.pi, a - 2 * math.pi, a))
    return a


def _split(obs):
    vel, goal = obs[:, :2], obs[:, 2:4]
    pts = obs[:, 4:].reshape(obs.shape[0], -1, 2)
    valid = ~((pts[:, :, 0] == 0) & (pts[:, :, 1] == 0))      # misses are exactly (0, 0)
    return vel, goal, pts, valid


def apf_act_batch(obs, a, w, k_att=50.0, k_rep=500.0, m=500.0, d0=10.0, n=2, min_vel=1.0):
    """APF_agent.act (APF.py:17-78) for obs [N, 26]; a, w = the robot's acceleration / angular-velocity
    tables (3 each).  Returns action indices [N] int64."""
    a = torch.as_tensor(a, dtype=obs.dtype, device=obs.device)
    w = torch.as_tensor(w, dtype=obs.dtype, device=obs.device)
    vel, goal, pts, valid = _split(obs)
    f_att = k_att * goal
    d_goal = torch.linalg.vector_norm(goal, dim=1)                                   # [N]
    d_obs = torch.linalg.vector_norm(pts, dim=2)                                     # [N, 11]
    d_safe = torch.where(valid, d_obs, torch.ones_like(d_obs))
    inv = 1.0 / d_safe - 1.0 / d0
    mag1 = k_rep * inv * (d_goal.unsqueeze(1) ** n) / (d_safe ** 2)                  # APF.py:37
    rep1 = mag1.unsqueeze(2) * (-pts / d_safe.unsqueeze(2))
    mag2 = (n / 2) * k_rep * inv ** 2 * (d_goal.unsqueeze(1) ** (n - 1))             # APF.py:42
    rep2 = mag2.unsqueeze(2) * (-goal / d_goal.unsqueeze(1)).unsqueeze(1)
    f_rep = torch.where(valid.unsqueeze(2), rep1 + rep2, torch.zeros_like(rep1)).sum(dim=1)
    f_tot = f_att + f_rep
    speed = torch.linalg.vector_norm(vel, dim=1)
    moving = speed > 1e-03
    v_angle = torch.where(moving, torch.atan2(vel[:, 1], vel[:, 0]), torch.zeros_like(speed))
    diff = _wrap_to_pi(torch.atan2(f_tot[:, 1], f_tot[:, 0]) - v_angle)
    w_idx = (w.unsqueeze(0) - diff.unsqueeze(1)).abs().argmin(dim=1)                 # APF.py:61
    v_dir = torch.where(moving.unsqueeze(1), vel / speed.clamp_min(1e-30).unsqueeze(1),
                        torch.tensor([1.0, 0.0], dtype=obs.dtype, device=obs.device).expand_as(vel))
    a_proj = ((f_tot / m) * v_dir).sum(dim=1)
    a_tab = a.unsqueeze(0).expand(obs.shape[0], -1)
    slow = (speed < min_vel).unsqueeze(1)
    a_eff = torch.where(slow & (a_tab <= 0.0), torch.full_like(a_tab, -float("inf")), a_tab)   # APF.py:71-74
    a_idx = (a_eff - a_proj.unsqueeze(1)).abs().argmin(dim=1)
    return a_idx * w.numel() + w_idx


def ba_act_batch(obs, a, w, follow_dist=5.0, detect_angle=2 * math.pi / 3, angle_margin=10 * math.pi / 180,
                 min_vel=1.0):
    """BA_agent.act (BA.py:14-155) for obs [N, 26].  Returns action indices [N] int64."""
    a = torch.as_tensor(a, dtype=obs.dtype, device=obs.device)
    w = torch.as_tensor(w, dtype=obs.dtype, device=obs.device)
    N = obs.shape[0]
    vel, goal, pts, valid = _split(obs)
    px, py = pts[:, :, 0], pts[:, :, 1]
    cnt = valid.sum(dim=1)
    ninf = torch.full_like(px, -float("inf"))
    ang = torch.atan2(py, px)
    max_angle = torch.where(valid, ang, ninf).max(dim=1).values
    min_angle = torch.where(valid, ang, -ninf).min(dim=1).values

    # move_to_goal (BA.py:73-84)
    speed = torch.linalg.vector_norm(vel, dim=1)
    g_angle = torch.atan2(goal[:, 1], goal[:, 0])
    v_mtg = torch.where((speed < 1e-03).unsqueeze(1), torch.tensor([1.0, 0.0], dtype=obs.dtype, device=obs.device).expand_as(vel), vel)
    diff_mtg = _wrap_to_pi(g_angle - torch.atan2(v_mtg[:, 1], v_mtg[:, 0]))
    w_mtg = (w.unsqueeze(0) - diff_mtg.unsqueeze(1)).abs().argmin(dim=1)
    a_mtg = torch.full((N,), int(a.argmax()), dtype=torch.int64, device=obs.device)

    # obstacle span (BA.py:52-65)
    hi = _wrap_to_pi(max_angle + angle_margin)
    lo = _wrap_to_pi(min_angle - angle_margin)
    hi = torch.where(hi >= 0.5 * detect_angle, torch.full_like(hi, math.pi), hi)
    lo = torch.where(lo <= -0.5 * detect_angle, torch.full_like(lo, -math.pi), lo)
    clear = (g_angle < lo) | (g_angle > hi)
    use_goal = (cnt == 0) | clear

    # wall_follow (BA.py:86-155): tangent `dir` and distance `d` by number of returns
    vf = valid.to(obs.dtype)
    order = torch.argsort((~valid).to(torch.int8), dim=1, stable=True)          # valid returns first, beam order kept
    p0 = torch.gather(pts, 1, order[:, :1].unsqueeze(2).expand(-1, -1, 2))[:, 0]
    p1 = torch.gather(pts, 1, order[:, 1:2].unsqueeze(2).expand(-1, -1, 2))[:, 0]
    # one return: d = |(x, 1)|, dir = R(90deg) (x, 1)   (sic: the reference uses the regression row [x, 1])
    d_1 = torch.sqrt(p0[:, 0] ** 2 + 1.0)
    dir_1 = torch.stack((-torch.ones_like(d_1), p0[:, 0]), dim=1)
    # two returns: rows [x0, 1], [x1, 1]  ->  dir = row1 - row0 = (x1 - x0, 0), v_1 = (x0, 1)
    dir_2 = torch.stack((p1[:, 0] - p0[:, 0], torch.zeros_like(d_1)), dim=1)
    cross_2 = (p0[:, 0] * dir_2[:, 1] - 1.0 * dir_2[:, 0]).abs()
    d_2 = cross_2 / torch.linalg.vector_norm(dir_2, dim=1)
    # three or more: normal equations of y = k x + c on the valid returns
    sxx = (vf * px * px).sum(1); sx = (vf * px).sum(1); sn_ = cnt.to(obs.dtype)
    sxy = (vf * px * py).sum(1); sy = (vf * py).sum(1)
    tr, det = sxx + sn_, sxx * sn_ - sx * sx
    disc = torch.sqrt((tr * tr / 4 - det).clamp_min(0.0))
    s0, s1 = tr / 2 + disc, tr / 2 - disc                                        # singular values of the PSD 2x2 A^T A
    vertical = s1 < 1e-03 * s0
    det_safe = torch.where(det == 0, torch.ones_like(det), det)
    k_ = (sn_ * sxy - sx * sy) / det_safe
    c_ = (-sx * sxy + sxx * sy) / det_safe
    dir_3 = torch.stack((torch.ones_like(k_), k_), dim=1)
    cross_3 = (1.0 * k_ - (k_ + c_) * 1.0).abs()                                 # v_1 = (1, k + c), dir = (1, k)
    d_3 = cross_3 / torch.linalg.vector_norm(dir_3, dim=1)
    dir_v = torch.stack((torch.zeros_like(k_), torch.ones_like(k_)), dim=1)
    d_v = (sx / sn_.clamp_min(1.0)).abs()
    dir_3 = torch.where(vertical.unsqueeze(1), dir_v, dir_3)
    d_3 = torch.where(vertical, d_v, d_3)
    one, two = (cnt == 1), (cnt == 2)
    wdir = torch.where(one.unsqueeze(1), dir_1, torch.where(two.unsqueeze(1), dir_2, dir_3))
    d = torch.where(one, d_1, torch.where(two, d_2, d_3))
    flip = (wdir * vel).sum(dim=1) < 0
    wdir = torch.where(flip.unsqueeze(1), -wdir, wdir)
    diff_wf = _wrap_to_pi(torch.atan2(wdir[:, 1], wdir[:, 0]) - torch.atan2(vel[:, 1], vel[:, 0]))
    close = d < follow_dist
    w_close = torch.where(diff_wf > 0, torch.full((N,), int(w.argmax()), device=obs.device),
                          torch.full((N,), int(w.argmin()), device=obs.device))
    w_wf = torch.where(close, w_close, (w.unsqueeze(0) - diff_wf.unsqueeze(1)).abs().argmin(dim=1))
    a_pos = torch.where(a > 0.0, a, torch.full_like(a, float("inf")))
    a_wf = torch.where(speed < min_vel, torch.full((N,), int(a_pos.argmin()), device=obs.device),
                       torch.full((N,), int(a.abs().argmin()), device=obs.device))

    w_idx = torch.where(use_goal, w_mtg, w_wf)
    a_idx = torch.where(use_goal, a_mtg, a_wf)
    return a_idx * w.numel() + w_idx
