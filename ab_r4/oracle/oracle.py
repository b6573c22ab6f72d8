"""ctypes wrapper around oracle/liboracle.so -- TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
The product package (distributional_rl_navigation_amd) never does.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

INFO_STRINGS = ("normal", "out of boundary", "too long episode", "collision", "reach goal")
MAXC, MAXO = 64, 64


def build(force=False):
    so = os.path.join(_HERE, "liboracle.so")
    src = os.path.join(_HERE, "marinenav_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", _HERE, "-B", "liboracle.so"], stdout=subprocess.DEVNULL)
    return so


def lib():
    global _LIB
    if _LIB is None:
        so = os.path.join(_HERE, "liboracle.so")
        if not os.path.exists(so):
            build()
        L = C.CDLL(so)
        dp, ip = C.POINTER(C.c_double), C.POINTER(C.c_int)
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_uint32]
        L.orc_destroy.argtypes = [C.c_void_p]
        L.orc_seed.argtypes = [C.c_void_p, C.c_uint32]
        L.orc_set_world_size.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_double]
        L.orc_set_schedule.argtypes = [C.c_void_p, C.c_int, C.POINTER(C.c_int64), ip, ip, dp]
        L.orc_set_flags.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        L.orc_set_start_goal.argtypes = [C.c_void_p] + [C.c_double] * 4
        L.orc_set_robot_N.argtypes = [C.c_void_p, C.c_int]
        L.orc_set_init_pose.argtypes = [C.c_void_p, C.c_double, C.c_double]
        L.orc_set_total_timesteps.argtypes = [C.c_void_p, C.c_int64]
        L.orc_get_velocity.argtypes = [C.c_void_p, C.c_double, C.c_double, dp]
        L.orc_get_observation.argtypes = [C.c_void_p, dp]
        L.orc_reset.argtypes = [C.c_void_p, dp]
        L.orc_load_world.argtypes = [C.c_void_p, C.c_int, dp, ip, dp, C.c_int, dp, dp, dp, dp, C.c_double, C.c_double, dp]
        L.orc_step.restype = C.c_int
        L.orc_step.argtypes = [C.c_void_p, C.c_int, dp, dp, ip]
        L.orc_get_state.argtypes = [C.c_void_p, dp, C.POINTER(C.c_int64)]
        L.orc_set_state.argtypes = [C.c_void_p, dp, C.c_int64]
        L.orc_get_world.argtypes = [C.c_void_p, dp, dp, ip, dp, dp]
        L.orc_peek_next_double.restype = C.c_double
        L.orc_peek_next_double.argtypes = [C.c_void_p]
        L.orc_rng_pos.restype = C.c_int
        L.orc_rng_pos.argtypes = [C.c_void_p]
        L.orc_rng_key.argtypes = [C.c_void_p, C.POINTER(C.c_uint32)]
        L.orc_rollout.restype = C.c_int64
        L.orc_rollout.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.c_int64, dp]
        _LIB = L
    return _LIB


def _dp(a):
    return a.ctypes.data_as(C.POINTER(C.c_double))


class OracleEnv:
    """One scalar float64 environment (mirrors marinenav_env.py MarineNavEnv)."""

    def __init__(self, seed=0, schedule=None):
        self.L = lib()
        self.h = C.c_void_p(self.L.orc_create(seed))
        if schedule is not None:
            self.set_schedule(schedule)

    def __del__(self):
        try:
            self.L.orc_destroy(self.h)
        except Exception:
            pass

    def seed(self, seed):
        self.L.orc_seed(self.h, seed)

    def set_world_size(self, nc, no, md):
        self.L.orc_set_world_size(self.h, int(nc), int(no), float(md))

    def set_schedule(self, schedule):
        ts = np.asarray(schedule["timesteps"], dtype=np.int64)
        nc = np.asarray(schedule["num_cores"], dtype=np.int32)
        no = np.asarray(schedule["num_obstacles"], dtype=np.int32)
        md = np.asarray(schedule["min_start_goal_dis"], dtype=np.float64)
        self.L.orc_set_schedule(self.h, len(ts), ts.ctypes.data_as(C.POINTER(C.c_int64)),
                                nc.ctypes.data_as(C.POINTER(C.c_int)), no.ctypes.data_as(C.POINTER(C.c_int)), _dp(md))

    def set_flags(self, reset_start_and_goal=True, random_reset_state=True, set_boundary=False):
        self.L.orc_set_flags(self.h, int(reset_start_and_goal), int(random_reset_state), int(set_boundary))

    def set_start_goal(self, start, goal):
        self.L.orc_set_start_goal(self.h, float(start[0]), float(start[1]), float(goal[0]), float(goal[1]))

    def set_robot_N(self, n):
        self.L.orc_set_robot_N(self.h, int(n))

    def set_total_timesteps(self, t):
        self.L.orc_set_total_timesteps(self.h, int(t))

    def reset(self):
        obs = np.zeros(26)
        self.L.orc_reset(self.h, _dp(obs))
        return obs

    def step(self, action):
        obs = np.zeros(26)
        r = C.c_double()
        info = C.c_int()
        done = self.L.orc_step(self.h, int(action), _dp(obs), C.byref(r), C.byref(info))
        return obs, r.value, bool(done), info.value

    def get_observation(self):
        obs = np.zeros(26)
        self.L.orc_get_observation(self.h, _dp(obs))
        return obs

    def get_velocity(self, x, y):
        v = np.zeros(2)
        self.L.orc_get_velocity(self.h, float(x), float(y), _dp(v))
        return v

    def load_world(self, cores, n_cores, obstacles, n_obs, start, goal, init_theta, init_speed):
        """cores: [n,4] = x, y, clockwise, Gamma ; obstacles: [n,3] = x, y, r"""
        cores = np.asarray(cores, dtype=np.float64)[:n_cores].reshape(-1, 4)
        obstacles = np.asarray(obstacles, dtype=np.float64)[:n_obs].reshape(-1, 3)
        cxy = np.ascontiguousarray(cores[:, :2]).ravel()
        cw = np.ascontiguousarray(cores[:, 2]).astype(np.int32)
        gm = np.ascontiguousarray(cores[:, 3])
        oxy = np.ascontiguousarray(obstacles[:, :2]).ravel()
        orad = np.ascontiguousarray(obstacles[:, 2])
        st = np.asarray(start, dtype=np.float64)
        gl = np.asarray(goal, dtype=np.float64)
        obs = np.zeros(26)
        self.L.orc_load_world(self.h, int(n_cores), _dp(cxy), cw.ctypes.data_as(C.POINTER(C.c_int)), _dp(gm),
                              int(n_obs), _dp(oxy), _dp(orad), _dp(st), _dp(gl), float(init_theta), float(init_speed), _dp(obs))
        return obs

    def load_eval_config(self, cfg):
        """marinenav_env.py:467-555 for the world/pose fields of one eval_config entry."""
        e, r = cfg["env"], cfg["robot"]
        nc, no = len(e["cores"]["positions"]), len(e["obstacles"]["positions"])
        cores = np.zeros((nc, 4))
        if nc:
            cores[:, :2] = e["cores"]["positions"]
            cores[:, 2] = e["cores"]["clockwise"]
            cores[:, 3] = e["cores"]["Gamma"]
        obst = np.zeros((no, 3))
        if no:
            obst[:, :2] = e["obstacles"]["positions"]
            obst[:, 2] = e["obstacles"]["r"]
        self.set_robot_N(r["N"])
        return self.load_world(cores, nc, obst, no, e["start"], e["goal"], r["init_theta"], r["init_speed"])

    def get_state(self):
        s = np.zeros(6)
        c = (C.c_int64 * 2)()
        self.L.orc_get_state(self.h, _dp(s), c)
        return s, int(c[0]), int(c[1])

    def set_state(self, s6, episode_timesteps=0):
        s = np.ascontiguousarray(s6, dtype=np.float64)
        self.L.orc_set_state(self.h, _dp(s), int(episode_timesteps))

    def get_world(self):
        cores = np.zeros((MAXC, 4))
        obst = np.zeros((MAXO, 3))
        meta = (C.c_int * 2)()
        sg = np.zeros(4)
        init = np.zeros(2)
        self.L.orc_get_world(self.h, _dp(cores), _dp(obst), meta, _dp(sg), _dp(init))
        return dict(cores=cores[: meta[0]], obstacles=obst[: meta[1]], n_cores=meta[0], n_obs=meta[1],
                    start=sg[:2].copy(), goal=sg[2:].copy(), init_theta=init[0], init_speed=init[1])

    def peek_next_double(self):
        return self.L.orc_peek_next_double(self.h)

    def rng_state(self):
        key = np.zeros(624, dtype=np.uint32)
        self.L.orc_rng_key(self.h, key.ctypes.data_as(C.POINTER(C.c_uint32)))
        return key, self.L.orc_rng_pos(self.h)

    def rollout(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        cs = C.c_double()
        ep = self.L.orc_rollout(self.h, a.ctypes.data_as(C.POINTER(C.c_int32)), len(a), C.byref(cs))
        return int(ep), cs.value
