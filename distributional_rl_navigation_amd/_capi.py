"""ctypes binding of libmarinenav_hip.so (include/marinenav_hip.h).

There is no CPU fallback: if the gfx950 library is missing or no GPU is visible the product path
raises.  The oracle under /oracle is test infrastructure and is never imported from here.
"""
import ctypes as C
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libmarinenav_hip.so")

MAX_CORES, MAX_OBS, NUM_BEAMS, OBS_DIM, NUM_ACTIONS, MAX_STAGES = 8, 10, 11, 26, 9, 8
PRECISION_F64, PRECISION_MIXED = 0, 1
INFO_STRINGS = ("normal", "out of boundary", "too long episode", "collision", "reach goal")


class MnParams(C.Structure):
    _fields_ = [
        ("width", C.c_double), ("height", C.c_double), ("core_r", C.c_double), ("v_rel_max", C.c_double),
        ("p", C.c_double), ("v_range", C.c_double * 2), ("obs_r_range", C.c_double * 2), ("clear_r", C.c_double),
        ("goal_dis", C.c_double), ("timestep_penalty", C.c_double), ("collision_penalty", C.c_double),
        ("goal_reward", C.c_double), ("discount", C.c_double), ("min_start_goal_dis", C.c_double),
        ("init_theta", C.c_double), ("init_speed", C.c_double), ("dt", C.c_double), ("robot_r", C.c_double),
        ("max_speed", C.c_double), ("a", C.c_double * 3), ("w", C.c_double * 3), ("sonar_range", C.c_double),
        ("sonar_angle", C.c_double), ("num_cores", C.c_int32), ("num_obs", C.c_int32),
        ("reset_start_and_goal", C.c_int32), ("random_reset_state", C.c_int32), ("set_boundary", C.c_int32),
        ("max_episode_steps", C.c_int32), ("N", C.c_int32), ("num_beams", C.c_int32), ("precision", C.c_int32),
        ("step_lanes", C.c_int32), ("rollout_lanes", C.c_int32),
    ]


class MarineNavHipError(RuntimeError):
    pass


def build(force=False):
    """Compile libmarinenav_hip.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    csrc = os.path.join(_HERE, "csrc")
    cmd = ["make", "-C", csrc] + (["-B"] if force else [])
    subprocess.check_call(cmd, stdout=subprocess.DEVNULL)
    return LIB_PATH


_lib = None

# (name, restype, argtypes) for every symbol include/marinenav_hip.h declares
_vp, _i32, _i64, _dbl = C.c_void_p, C.c_int32, C.c_int64, C.c_double
_pd, _pi32, _pi64, _pu8, _pu32, _pf = (C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_int64),
                                         C.c_void_p, C.POINTER(C.c_uint32), C.c_void_p)
SIGNATURES = [
    ("mn_default_params", C.c_int, [C.POINTER(MnParams)]),
    ("mn_create", C.c_int, [_i32, C.POINTER(MnParams), C.POINTER(_vp)]),
    ("mn_destroy", C.c_int, [_vp]),
    ("mn_last_error", C.c_char_p, [_vp]),
    ("mn_num_envs", _i32, [_vp]),
    ("mn_set_params", C.c_int, [_vp, C.POINTER(MnParams)]),
    ("mn_get_params", C.c_int, [_vp, C.POINTER(MnParams)]),
    ("mn_seed", C.c_int, [_vp, _pu32, _vp]),
    ("mn_set_schedule", C.c_int, [_vp, _i32, _pi64, _pi32, _pi32, _pd, _dbl]),
    ("mn_set_start_goal", C.c_int, [_vp, _i32, _pd, _pd]),
    ("mn_reset", C.c_int, [_vp, _pu8, _pf, _vp]),
    ("mn_step", C.c_int, [_vp, _vp, _pf, _pf, _pu8, _pu8, _vp]),
    ("mn_step_append", C.c_int, [_vp, _vp, _pf, _pf, _pf, _pu8, _pu8, _pf, _pf, _vp, _pf, _pf, _i64, _i64, _vp]),
    ("mn_build_info", _i32, []),
    ("mn_rollout", C.c_int, [_vp, _i32, _vp, C.c_uint64, C.c_uint64, C.c_uint64, _pf, _pf, _pf, _pu8, _pu8, _vp, _vp]),
    ("mn_random_actions", C.c_int, [C.c_uint64, C.c_uint64, C.c_uint64, _i32, _vp, _vp]),
    ("mn_reset_done", C.c_int, [_vp, _pf, _vp]),
    ("mn_reset_done_async", C.c_int, [_vp, _pf, _vp, C.POINTER(_vp), C.POINTER(C.c_uint32)]),
    ("mn_reset_join", C.c_int, [_vp, _vp]),
    ("mn_set_reset_under_act_max", C.c_int, [_vp, _i32, C.POINTER(C.c_int64)]),
    ("mn_debug_side_delay_us", C.c_int, [_vp, _i32]),
    ("mn_load_worlds", C.c_int, [_vp, _i32, _i32, _pi32, _pd, _pi32, _pd, _pi32, _pd, _pd, _pd, _pd, _pd, _pd, _pf, _vp]),
    ("mn_get_worlds", C.c_int, [_vp, _i32, _i32, _pi32, _pd, _pi32, _pd, _pi32, _pd, _pd, _pd, _pd, _pd, _pd]),
    ("mn_get_state", C.c_int, [_vp, _i32, _i32, _pd, _pi32, _pi64]),
    ("mn_set_state", C.c_int, [_vp, _i32, _i32, _pd, _pi32, _pi64]),
    ("mn_enable_obs64", C.c_int, [_vp, _i32]),
    ("mn_get_obs64", C.c_int, [_vp, _i32, _i32, _pd]),
    ("mn_get_reward64", C.c_int, [_vp, _i32, _i32, _pd]),
    ("mn_enable_trajectory", C.c_int, [_vp, _i32]),
    ("mn_get_trajectory", C.c_int, [_vp, _i32, _i32, _i32, _pd]),
    ("mn_peek_next_double", C.c_int, [_vp, _i32, _i32, _pd]),
    ("mn_last_done_count", C.c_int, [_vp, _vp, _pi32]),
    ("mn_profile_begin", C.c_int, [_vp, _i32]),
    ("mn_profile_end", C.c_int, [_vp, _vp, _pd, _pi32]),
    ("mn_profile_reset_end", C.c_int, [_vp, _vp, _pd, _pi32]),
    ("mn_iqn_create", C.c_int, [C.POINTER(_vp)]),
    ("mn_iqn_destroy", C.c_int, [_vp]),
    ("mn_iqn_weights_changed", C.c_int, [_vp]),
    ("mn_iqn_set_variant", C.c_int, [_vp, _i32]),
    ("mn_iqn_set_grid", C.c_int, [_vp, _i32]),
    ("mn_iqn_set_tau_mode", C.c_int, [_vp, _i32]),
    ("mn_iqn_set_late_rows", C.c_int, [_vp, _vp, _vp, C.c_uint32, _i32]),
    ("mn_iqn_late_timeouts", C.c_int, [_vp, _vp, C.POINTER(C.c_uint32)]),
    ("mn_iqn_late_timeouts_peek", C.c_int, [_vp, C.POINTER(C.c_uint32)]),
    ("mn_iqn_set_late_bound_ms", C.c_int, [_vp, C.c_double]),
    ("mn_iqn_train_workspace_init", C.c_int, [_vp, _i32, _vp]),
    ("mn_iqn_train_step", C.c_int, [_vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float, _i32,
                                    _dbl, _dbl, _dbl, _dbl, _dbl, _vp]),
    ("mn_rollout_policy", C.c_int, [_vp, _i32, _i32, _vp, _vp, _vp, _vp, _vp, _vp, _vp]),
    ("mn_planner_act", C.c_int, [_vp, _i32, _i32, _pd, _pd, _vp, _vp]),
    ("mn_dqn_image_floats", C.c_int64, []),
    ("mn_dqn_act", C.c_int, [_vp, C.POINTER(C.c_void_p), _vp, _i32, _vp, _vp, _i32, _vp]),
    ("mn_xchg_create", C.c_int, [_i32, _i32, C.POINTER(_vp)]),
    ("mn_xchg_export", C.c_int, [_vp, _vp]),
    ("mn_xchg_import", C.c_int, [_vp, _i32, _vp]),
    ("mn_xchg_attach", C.c_int, [_vp, _vp, _i32, _vp]),
    ("mn_iqn_train_exchange", C.c_int, [_vp, _vp, _vp, _i32, C.c_float, _vp]),
    ("mn_xchg_memory_kind", C.c_int, [_vp]),
    ("mn_xchg_set_timeout_ms", C.c_int, [_vp, _i64]),
    ("mn_iqn_train_plan", C.c_int, [_i32, _i32, _i32]),
    ("mn_iqn_train_set_cu_limit", C.c_int, [_i32]),
    ("mn_iqn_train_workspace_status_word", C.c_int64, [_i32]),
    ("mn_iqn_train_step_xchg", C.c_int, [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i32, _i32, C.c_float,
                                         _i32, _dbl, _dbl, _dbl, _dbl, _dbl, C.c_float, _vp]),
    ("mn_xchg_status", C.c_int, [_vp, _pi32]),
    ("mn_xchg_last_error", C.c_char_p, [_vp]),
    ("mn_xchg_destroy", C.c_int, [_vp]),
    ("mn_probe_mfma_clock", C.c_int, [C.c_double, _pd, _vp]),
    ("mn_iqn_refresh", C.c_int, [_vp, C.POINTER(C.c_void_p), _vp]),
    ("mn_iqn_act", C.c_int, [_vp, _vp, _vp, C.POINTER(C.c_void_p), _vp, _vp, C.c_float, _vp, _vp, _i32, _i32, _vp]),
    ("mn_iqn_act_rng", C.c_int, [_vp, _vp, C.POINTER(C.c_void_p), _vp, _vp, _vp, C.c_float, C.c_float, _vp, _vp, _vp, _i32, _i32, _vp]),
    ("mn_replay_append", C.c_int, [_vp] * 10 + [_i64, _i64, _i64, _vp]),
    ("mn_iqn_train_workspace_floats", C.c_int64, [_i32]),
    ("mn_iqn_train_workspace_misplaced_word", C.c_int64, [_i32]),
    ("mn_iqn_sample", C.c_int, [_i64, _i32, _vp, _vp, _vp, _i32, _vp]),
    ("mn_iqn_train_grad", C.c_int, [_vp] * 13 + [_i32, _i32, C.c_float, _vp]),
    ("mn_iqn_train_grad_sampled", C.c_int, [_vp] * 5 + [_i64] + [_vp] * 8 + [_i32, _i32, C.c_float, _i32, _vp]),
    ("mn_iqn_train_adam", C.c_int, [_vp] * 6 + [_i32] + [C.c_double] * 5 + [C.c_float, _i32, _vp]),
    ("mn_iqn_train_set_mode", C.c_int, [_i32]),
    ("mn_iqn_profile_begin", C.c_int, [_vp, _i32]),
    ("mn_iqn_profile_end", C.c_int, [_vp, _vp, _pd, _pi32]),
]


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise MarineNavHipError(
                f"{LIB_PATH} not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                "(hipcc --offload-arch=gfx950).  There is no CPU fallback.")
        L = C.CDLL(LIB_PATH)
        for name, res, args in SIGNATURES:
            fn = getattr(L, name)      # AttributeError if the header and the library disagree
            fn.restype, fn.argtypes = res, args
        if L.mn_build_info() != 0:
            raise MarineNavHipError(f"{LIB_PATH} is an ablation build (mn_build_info() != 0); the package only runs the full kernels")
        _lib = L
    return _lib


def default_params():
    p = MnParams()
    rc = lib().mn_default_params(C.byref(p))
    if rc:
        raise MarineNavHipError("mn_default_params failed")
    return p


def check(rc, handle=None):
    if rc != 0:
        msg = lib().mn_last_error(handle)
        raise MarineNavHipError(f"libmarinenav_hip error {rc}: {msg.decode() if msg else ''}")


def stream_ptr(device):
    """The calling thread's current HIP stream on `device` as a ctypes pointer -- what every per-step entry point takes.  torch.cuda.current_stream() builds a Stream
    object per call (~4 us, five times per vector step); the raw accessor returns the handle itself."""
    import torch
    idx = device.index if device.index is not None else torch.cuda.current_device()
    try:
        return C.c_void_p(torch._C._cuda_getCurrentRawStream(idx))
    except AttributeError:      # (another torch build)
        return C.c_void_p(torch.cuda.current_stream(device).cuda_stream)

