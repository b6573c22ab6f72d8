"""Batched marinenav_env on one MI355X: host mirror of the reference's MarineNavEnv
(marinenav_env/envs/marinenav_env.py) over the C-ABI in include/marinenav_hip.h.

All per-step data stays in HBM as torch tensors; this class only owns buffers and forwards raw
device pointers + the current HIP stream to the hand-written gfx950 kernels.
"""
import ctypes as C

import numpy as np
import torch

from .. import _capi
from .._capi import MAX_CORES, MAX_OBS, OBS_DIM, INFO_STRINGS

_ATTR_TO_PARAM = {
    # MarineNavEnv attribute (marinenav_env.py:40-73) -> mn_params field
    "width": "width", "height": "height", "r": "core_r", "v_rel_max": "v_rel_max", "p": "p",
    "clear_r": "clear_r", "goal_dis": "goal_dis", "timestep_penalty": "timestep_penalty",
    "collision_penalty": "collision_penalty", "goal_reward": "goal_reward", "discount": "discount",
    "num_cores": "num_cores", "num_obs": "num_obs", "min_start_goal_dis": "min_start_goal_dis",
    "reset_start_and_goal": "reset_start_and_goal", "random_reset_state": "random_reset_state",
    "set_boundary": "set_boundary", "init_theta": "init_theta", "init_speed": "init_speed",
}


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _np_ptr(a, ctype):
    return a.ctypes.data_as(C.POINTER(ctype))


def shard_seeds(n_envs, seed=0, first_index=0):
    """Default per-env seeds of a shard: env i of the shard is global env `first_index + i` and is seeded with
    `seed + first_index + i` (mod 2^32, np.random.RandomState's seed range), so rank r of a multi-GPU run
    (first_index = r * n_envs) generates exactly the worlds of rows [r n, (r+1) n) of a one-GPU run."""
    if n_envs <= 0 or first_index < 0:
        raise ValueError("shard_seeds: n_envs must be positive and first_index non-negative")
    return ((np.arange(int(n_envs), dtype=np.uint64) + np.uint64(int(seed) % (1 << 32)) + np.uint64(int(first_index)))
            & np.uint64(0xFFFFFFFF)).astype(np.uint32)


class VecMarineNavEnv:
    """n_envs independent MarineNavEnv instances stepped by one kernel launch.

    Env i is seeded with ``seeds[i]`` (default ``seed + first_index + i``) exactly like
    ``MarineNavEnv(seed=...)`` (marinenav_env.py:27,75-78): the same seed reproduces the
    reference's worlds bit for bit.
    """

    def __init__(self, n_envs, seed=0, seeds=None, schedule=None, device="cuda:0", precision="mixed",
                 timestep_scale=1.0, first_index=0, params=None, step_lanes=0, rollout_lanes=0, obs64=False):
        if not torch.cuda.is_available():
            raise _capi.MarineNavHipError("VecMarineNavEnv needs a ROCm GPU (MI355X); there is no CPU fallback")
        self.L = _capi.lib()
        self.device = torch.device(device)
        if self.device.type != "cuda":
            raise _capi.MarineNavHipError(f"VecMarineNavEnv(device={device!r}): the env kernels are HIP kernels for gfx950 -- there is no CPU path (the reference's "
                                          "`-D cpu` has no counterpart here; the CPU restatement under oracle/ is test infrastructure)")
        torch.cuda.set_device(self.device)
        self.n_envs = int(n_envs)
        self.params = params if params is not None else _capi.default_params()
        self.params.precision = _capi.PRECISION_F64 if precision in ("f64", "float64", 0) else _capi.PRECISION_MIXED
        self.precision = "f64" if self.params.precision == _capi.PRECISION_F64 else "mixed"
        self.params.step_lanes = int(step_lanes)
        self.params.rollout_lanes = int(rollout_lanes)
        h = C.c_void_p()
        rc = self.L.mn_create(self.n_envs, C.byref(self.params), C.byref(h))
        if rc:
            raise _capi.MarineNavHipError(f"mn_create failed ({rc}): {self.L.mn_last_error(None).decode()}")
        self.h = h
        self.first_index = int(first_index)
        self.obs64_enabled = False
        if obs64:      # float64 copies of observations / rewards (parity checks, the n = 1 facade); off in the training loop
            self.enable_obs64(True)
        if seeds is None:
            seeds = shard_seeds(self.n_envs, seed, first_index)
        self.seed(seeds)
        self.schedule = None
        if schedule is not None:
            self.set_schedule(schedule, timestep_scale)
        dev = self.device
        # observations are double-buffered: step() writes the half that does NOT hold the
        # observations returned by the previous step()/reset(), so (obs_t, obs_t+1) are both
        # resident for the replay append without a copy
        self._obs_bufs = [torch.zeros(self.n_envs, OBS_DIM, dtype=torch.float32, device=dev) for _ in range(2)]
        self._cur = 0
        self.obs = self._obs_bufs[0]
        self.reward = torch.zeros(self.n_envs, dtype=torch.float32, device=dev)
        self.done = torch.zeros(self.n_envs, dtype=torch.uint8, device=dev)
        self.info = torch.zeros(self.n_envs, dtype=torch.uint8, device=dev)
        self._terminal_obs = None
        self.late_rows = None      # reset_done(under_next_act=True)
        self.reset_under_act_max = self.RESET_UNDER_ACT_MAX_DEFAULT
        self.reset_launches = [0, 0]      # ... how many of those calls the library ran [in front of, under] the next act kernel

    # ---- lifecycle ---------------------------------------------------------------------------
    def close(self):
        if getattr(self, "h", None) is not None and self.h:
            torch.cuda.synchronize(self.device)
            self.L.mn_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _stream(self):
        return _capi.stream_ptr(self.device)

    def _check(self, rc):
        _capi.check(rc, self.h)

    # ---- configuration -----------------------------------------------------------------------
    def seed(self, seeds):
        """MarineNavEnv.seed (marinenav_env.py:75-78) per env."""
        s = np.ascontiguousarray(np.broadcast_to(np.asarray(seeds, dtype=np.uint32), (self.n_envs,)))
        self._check(self.L.mn_seed(self.h, _np_ptr(s, C.c_uint32), self._stream()))
        self.seeds = s
        return list(s)

    def set_schedule(self, schedule, timestep_scale=1.0):
        """`schedule` ctor argument (marinenav_env.py:27,89-98); None clears it."""
        if schedule is None:
            self._check(self.L.mn_set_schedule(self.h, 0, None, None, None, None, 1.0))
            self.schedule = None
            return
        ts = np.ascontiguousarray(schedule["timesteps"], dtype=np.int64)
        nc = np.ascontiguousarray(schedule["num_cores"], dtype=np.int32)
        no = np.ascontiguousarray(schedule["num_obstacles"], dtype=np.int32)
        md = np.ascontiguousarray(schedule["min_start_goal_dis"], dtype=np.float64)
        self._check(self.L.mn_set_schedule(self.h, len(ts), _np_ptr(ts, C.c_int64), _np_ptr(nc, C.c_int32),
                                           _np_ptr(no, C.c_int32), _np_ptr(md, C.c_double), float(timestep_scale)))
        self.schedule = schedule

    def set_attrs(self, **kw):
        """Attribute writes of the reference (`env.num_cores = 4`, `env.robot.N = 5`, ...)."""
        for k, v in kw.items():
            if k in _ATTR_TO_PARAM:
                setattr(self.params, _ATTR_TO_PARAM[k], type(getattr(self.params, _ATTR_TO_PARAM[k]))(v))
            elif k in ("N", "dt", "max_speed", "robot_r", "sonar_range", "sonar_angle"):
                setattr(self.params, k, type(getattr(self.params, k))(v))
            elif k in ("v_range", "obs_r_range", "a", "w"):
                arr = getattr(self.params, k)
                for i, x in enumerate(v):
                    arr[i] = float(x)
            else:
                raise AttributeError(f"unknown env attribute {k}")
        self._check(self.L.mn_set_params(self.h, C.byref(self.params)))

    def set_start_goal(self, start, goal, env_idx=-1):
        s = (C.c_double * 2)(float(start[0]), float(start[1]))
        g = (C.c_double * 2)(float(goal[0]), float(goal[1]))
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_set_start_goal(self.h, int(env_idx), s, g))

    # ---- gym-shaped batched API --------------------------------------------------------------
    def get_state_space_dimension(self):
        return OBS_DIM

    def get_action_space_dimension(self):
        return _capi.NUM_ACTIONS

    @property
    def discount(self):
        return self.params.discount

    def reset(self, mask=None):
        """MarineNavEnv.reset for every env (or the masked ones).  Returns obs [n,26] f32 (device)."""
        m = None
        if mask is not None:
            m = mask.to(device=self.device, dtype=torch.uint8).contiguous()
        self._check(self.L.mn_reset(self.h, _ptr(m) if m is not None else None, _ptr(self.obs), self._stream()))
        return self.obs

    def step(self, actions):
        """MarineNavEnv.step for every env; no auto-reset (like the reference).
        Returns (obs [n,26] f32, reward [n] f32, done [n] u8, info [n] u8 code) -- device tensors
        owned by the env; reward/done/info are overwritten by the next call, obs by the one after."""
        self._cur ^= 1
        self.obs = self._obs_bufs[self._cur]
        a = actions
        if a.dtype != torch.int32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.int32).contiguous()
        self._check(self.L.mn_step(self.h, _ptr(a), _ptr(self.obs), _ptr(self.reward), _ptr(self.done), _ptr(self.info),
                                   self._stream()))
        return self.obs, self.reward, self.done, self.info

    def step_append(self, actions, prev_obs, replay):
        """`step` + `replay.add` for every env in ONE launch (C-ABI mn_step_append): the transition
        (prev_obs[i], actions[i], reward, obs, done) of env i is written straight into the device ring of `replay`
        (iqn.replay_buffer.ReplayBuffer) by the step kernel.  `prev_obs` must be the tensor the last step()/reset()
        returned (the other half of the observation double buffer)."""
        self._cur ^= 1
        self.obs = self._obs_bufs[self._cur]
        a = actions
        if a.dtype != torch.int32 or not a.is_contiguous() or a.device != self.device:
            a = a.to(device=self.device, dtype=torch.int32).contiguous()
        assert prev_obs.data_ptr() != self.obs.data_ptr() and prev_obs.is_contiguous() and prev_obs.dtype == torch.float32
        self._check(self.L.mn_step_append(self.h, _ptr(a), _ptr(prev_obs), _ptr(self.obs), _ptr(self.reward), _ptr(self.done),
                                          _ptr(self.info), _ptr(replay.states), _ptr(replay.next_states), _ptr(replay.actions),
                                          _ptr(replay.rewards), _ptr(replay.dones), int(replay.ptr), int(replay.capacity),
                                          self._stream()))
        replay.advance(self.n_envs)
        return self.obs, self.reward, self.done, self.info

    def rollout(self, n_steps, actions=None, action_seed=0, first_step=0, trace=("obs", "reward", "done")):
        """`n_steps` vector steps with auto-reset in ONE launch (C-ABI mn_rollout): uniformly random actions drawn inside
        the kernel (counter-based: seed, step index, global env index = first_index + i), or `actions` [n_steps, n] i32.
        Bit-identical to n_steps x (step, reset_done).  Returns a dict with the final `obs` [n,26] (what the next act
        would see) and the requested traces: obs [T,n,26], reward [T,n], done [T,n] u8, info [T,n] u8, action [T,n] i32."""
        T, n, dev = int(n_steps), self.n_envs, self.device
        key = (T, tuple(sorted(trace)))
        bufs = getattr(self, "_rollout_bufs", None)
        if bufs is None or bufs[0] != key:
            mk = dict(obs=lambda: torch.empty(T, n, OBS_DIM, dtype=torch.float32, device=dev),
                      reward=lambda: torch.empty(T, n, dtype=torch.float32, device=dev),
                      done=lambda: torch.empty(T, n, dtype=torch.uint8, device=dev),
                      info=lambda: torch.empty(T, n, dtype=torch.uint8, device=dev),
                      action=lambda: torch.empty(T, n, dtype=torch.int32, device=dev))
            bufs = self._rollout_bufs = (key, {k: mk[k]() for k in trace})
        tr = bufs[1]
        a = None
        if actions is not None:
            a = actions.to(device=dev, dtype=torch.int32).contiguous()
            assert a.shape == (T, n)
        p = lambda k: _ptr(tr[k]) if k in tr else None
        self._check(self.L.mn_rollout(self.h, T, _ptr(a) if a is not None else None, int(action_seed), int(first_step),
                                      int(self.first_index), _ptr(self.obs), p("obs"), p("reward"), p("done"), p("info"), p("action"),
                                      self._stream()))
        out = dict(tr)
        out["final_obs"] = self.obs
        return out

    POLICIES = {"APF": 1, "BA": 2}      # MN_POLICY_APF / MN_POLICY_BA (include/marinenav_hip.h)

    def rollout_policy(self, n_steps, policy, trace=("reward", "done", "info", "action")):
        """Every env's CURRENT episode under the classical baseline `policy` ("APF" = APF.py:17-78, "BA" = BA.py:14-155) for up to
        `n_steps` steps in ONE launch (C-ABI mn_rollout_policy): the policy is evaluated on the device on each step's observation row.
        No resets: a finished env idles (reward 0, done 1, terminal info, action -1 in the traces).  Step for step identical to the
        loop (planners.planner_act_batch, step).  Returns the requested traces + `final_obs` (terminal observations where finished)."""
        T, n, dev = int(n_steps), self.n_envs, self.device
        mk = dict(obs=lambda: torch.zeros(T, n, OBS_DIM, dtype=torch.float32, device=dev),
                  reward=lambda: torch.empty(T, n, dtype=torch.float32, device=dev),
                  done=lambda: torch.empty(T, n, dtype=torch.uint8, device=dev),
                  info=lambda: torch.empty(T, n, dtype=torch.uint8, device=dev),
                  action=lambda: torch.empty(T, n, dtype=torch.int32, device=dev))
        tr = {k: mk[k]() for k in trace}
        p = lambda k: _ptr(tr[k]) if k in tr else None
        self._check(self.L.mn_rollout_policy(self.h, T, int(self.POLICIES[policy]), _ptr(self.obs), p("obs"), p("reward"), p("done"), p("info"),
                                             p("action"), self._stream()))
        out = dict(tr)
        out["final_obs"] = self.obs
        return out

    def random_actions(self, action_seed, step):
        """The actions `rollout(action_seed=...)` takes at step index `step` (int32 [n] on the device)."""
        a = torch.empty(self.n_envs, dtype=torch.int32, device=self.device)
        self._check(self.L.mn_random_actions(int(action_seed), int(step), int(self.first_index), self.n_envs, _ptr(a), self._stream()))
        return a

    def reset_done(self, keep_terminal_obs=False, under_next_act=False):
        """The caller-side `if done: state = env.reset()` (agent.py:152-170) for the whole batch.
        Overwrites the rows of finished envs in ``self.obs`` with their first observation; with
        keep_terminal_obs the pre-reset observations are preserved in ``self.terminal_obs``.
        `under_next_act` (mn_reset_done_async): the reset runs on the handle's own stream, off the caller's critical path; ``self.late_rows``
        = (done flags, "row is final" words, tick) is what the next act launch needs to take the finished envs' rows last
        (`fused_act(..., late_rows=env.take_late_rows())`).  Until `join_reset()` -- or the next step / reset of this env -- the rows of
        finished envs in ``self.obs`` must not be read by anything else on the caller's stream."""
        if keep_terminal_obs:
            if self._terminal_obs is None:
                self._terminal_obs = torch.empty_like(self.obs)
            self._terminal_obs.copy_(self.obs)
        if under_next_act:
            ready, tick = C.c_void_p(), C.c_uint32()
            self._check(self.L.mn_reset_done_async(self.h, _ptr(self.obs), self._stream(), C.byref(ready), C.byref(tick)))
            # (the library runs it in front after all -- ready NULL -- while many episodes end per vector step: set_reset_under_act_max)
            self.late_rows = (self.done, ready.value, tick.value) if ready.value else None
            self.reset_launches[1 if ready.value else 0] += 1
        else:
            self.late_rows = None
            self._check(self.L.mn_reset_done(self.h, _ptr(self.obs), self._stream()))
        return self.obs

    RESET_UNDER_ACT_MAX_DEFAULT = 5000      # = MN_RESET_UNDER_ACT_MAX_DEFAULT (include/marinenav_hip.h)

    def set_reset_under_act_max(self, max_resets=None):
        """`reset_done(under_next_act=True)` goes under the act kernel only while the decaying peak of the episodes started per reset launch is at most this
        (None: the library's default, 5000; 2**31 - 1: always, -1: never).  Returns that peak as of the last launch seen (-1: none yet)."""
        last = C.c_int64()
        self.reset_under_act_max = self.RESET_UNDER_ACT_MAX_DEFAULT if max_resets is None else int(max_resets)
        self._check(self.L.mn_set_reset_under_act_max(self.h, self.reset_under_act_max, C.byref(last)))
        return last.value

    def debug_side_delay_us(self, us):
        """Test hook (mn_debug_side_delay_us): every reset launch under the act kernel is held back by `us` microseconds on the handle's own stream."""
        self._check(self.L.mn_debug_side_delay_us(self.h, int(us)))

    def take_late_rows(self):
        """The pending `reset_done(under_next_act=True)`'s late rows for ONE act launch (None if there is none)."""
        lr, self.late_rows = getattr(self, "late_rows", None), None
        return lr

    def join_reset(self):
        """Everything enqueued on the current stream from here on runs after a pending `reset_done(under_next_act=True)`."""
        self.late_rows = None
        self._check(self.L.mn_reset_join(self.h, self._stream()))

    @property
    def terminal_obs(self):
        return self._terminal_obs

    def step_autoreset(self, actions, keep_terminal_obs=False):
        """step + reset_done.  Returns (obs_for_next_act, reward, done, info)."""
        self.step(actions)
        self.reset_done(keep_terminal_obs)
        return self.obs, self.reward, self.done, self.info

    def last_done_count(self):
        out = C.c_int32()
        self._check(self.L.mn_last_done_count(self.h, self._stream(), C.byref(out)))
        return out.value

    # ---- worlds ------------------------------------------------------------------------------
    def load_worlds(self, worlds, first_env=0):
        """reset_with_eval_config (marinenav_env.py:467-555) for consecutive envs.  `worlds` is a
        list of dicts with keys cores [n,4]=(x,y,clockwise,Gamma), obstacles [n,3]=(x,y,r), start,
        goal, init_theta, init_speed.  Returns the first observations of those envs."""
        cnt = len(worlds)
        ncs = np.zeros(cnt, np.int32); nos = np.zeros(cnt, np.int32)
        cxy = np.zeros((cnt, MAX_CORES, 2)); cw = np.zeros((cnt, MAX_CORES), np.int32); gm = np.zeros((cnt, MAX_CORES))
        oxy = np.zeros((cnt, MAX_OBS, 2)); orr = np.zeros((cnt, MAX_OBS))
        st = np.zeros((cnt, 2)); gl = np.zeros((cnt, 2)); th = np.zeros(cnt); sp = np.zeros(cnt)
        for i, w in enumerate(worlds):
            c = np.asarray(w["cores"], dtype=np.float64).reshape(-1, 4)
            o = np.asarray(w["obstacles"], dtype=np.float64).reshape(-1, 3)
            if len(c) > MAX_CORES or len(o) > MAX_OBS:
                raise ValueError("world exceeds device capacity (8 cores, 10 obstacles)")
            ncs[i], nos[i] = len(c), len(o)
            cxy[i, :len(c)] = c[:, :2]; cw[i, :len(c)] = c[:, 2] != 0; gm[i, :len(c)] = c[:, 3]
            oxy[i, :len(o)] = o[:, :2]; orr[i, :len(o)] = o[:, 2]
            st[i] = w["start"]; gl[i] = w["goal"]; th[i] = w["init_theta"]; sp[i] = w["init_speed"]
        d = C.c_double
        self._check(self.L.mn_load_worlds(self.h, int(first_env), cnt, _np_ptr(ncs, C.c_int32), _np_ptr(cxy, d),
                                          _np_ptr(cw, C.c_int32), _np_ptr(gm, d), _np_ptr(nos, C.c_int32), _np_ptr(oxy, d),
                                          _np_ptr(orr, d), _np_ptr(st, d), _np_ptr(gl, d), _np_ptr(th, d), _np_ptr(sp, d),
                                          _ptr(self.obs), self._stream()))
        return self.obs[first_env:first_env + cnt]

    @staticmethod
    def world_from_eval_config(cfg):
        """One entry of eval_config.json (episode_data schema, marinenav_env.py:557-622) -> world dict."""
        e, r = cfg["env"], cfg["robot"]
        nc, no = len(e["cores"]["positions"]), len(e["obstacles"]["positions"])
        cores = np.zeros((nc, 4)); obst = np.zeros((no, 3))
        if nc:
            cores[:, :2] = e["cores"]["positions"]; cores[:, 2] = e["cores"]["clockwise"]; cores[:, 3] = e["cores"]["Gamma"]
        if no:
            obst[:, :2] = e["obstacles"]["positions"]; obst[:, 2] = e["obstacles"]["r"]
        return dict(cores=cores, obstacles=obst, start=e["start"], goal=e["goal"],
                    init_theta=r["init_theta"], init_speed=r["init_speed"])

    def get_worlds(self, first_env=0, count=None):
        cnt = self.n_envs - first_env if count is None else count
        ncs = np.zeros(cnt, np.int32); nos = np.zeros(cnt, np.int32)
        cxy = np.zeros((cnt, MAX_CORES, 2)); cw = np.zeros((cnt, MAX_CORES), np.int32); gm = np.zeros((cnt, MAX_CORES))
        oxy = np.zeros((cnt, MAX_OBS, 2)); orr = np.zeros((cnt, MAX_OBS))
        st = np.zeros((cnt, 2)); gl = np.zeros((cnt, 2)); th = np.zeros(cnt); sp = np.zeros(cnt)
        d = C.c_double
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_get_worlds(self.h, int(first_env), cnt, _np_ptr(ncs, C.c_int32), _np_ptr(cxy, d),
                                         _np_ptr(cw, C.c_int32), _np_ptr(gm, d), _np_ptr(nos, C.c_int32), _np_ptr(oxy, d),
                                         _np_ptr(orr, d), _np_ptr(st, d), _np_ptr(gl, d), _np_ptr(th, d), _np_ptr(sp, d)))
        out = []
        for i in range(cnt):
            c = np.concatenate([cxy[i, :ncs[i]], cw[i, :ncs[i], None].astype(np.float64), gm[i, :ncs[i], None]], axis=1)
            o = np.concatenate([oxy[i, :nos[i]], orr[i, :nos[i], None]], axis=1)
            out.append(dict(cores=c, obstacles=o, n_cores=int(ncs[i]), n_obs=int(nos[i]), start=st[i].copy(),
                            goal=gl[i].copy(), init_theta=float(th[i]), init_speed=float(sp[i])))
        return out

    # ---- state accessors (tests, checkpoints) ------------------------------------------------
    def get_state(self, first_env=0, count=None):
        cnt = self.n_envs - first_env if count is None else count
        s = np.zeros((cnt, 6)); ep = np.zeros(cnt, np.int32); tot = np.zeros(cnt, np.int64)
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_get_state(self.h, int(first_env), cnt, _np_ptr(s, C.c_double), _np_ptr(ep, C.c_int32),
                                        _np_ptr(tot, C.c_int64)))
        return s, ep, tot

    def set_state(self, state=None, episode_timesteps=None, total_timesteps=None, first_env=0):
        cnt = len(state) if state is not None else (len(episode_timesteps) if episode_timesteps is not None else len(total_timesteps))
        s = np.ascontiguousarray(state, dtype=np.float64) if state is not None else None
        ep = np.ascontiguousarray(episode_timesteps, dtype=np.int32) if episode_timesteps is not None else None
        tot = np.ascontiguousarray(total_timesteps, dtype=np.int64) if total_timesteps is not None else None
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_set_state(self.h, int(first_env), cnt,
                                        _np_ptr(s, C.c_double) if s is not None else None,
                                        _np_ptr(ep, C.c_int32) if ep is not None else None,
                                        _np_ptr(tot, C.c_int64) if tot is not None else None))

    def enable_obs64(self, on=True):
        """Keep float64 copies of every later step's / reset's observation rows and rewards (`get_obs64`, `get_reward64`; f64 handles).
        Off by default: the training loop never reads them (C-ABI mn_enable_obs64)."""
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_enable_obs64(self.h, 1 if on else 0))
        self.obs64_enabled = bool(on)

    def get_obs64(self, first_env=0, count=None):
        cnt = self.n_envs - first_env if count is None else count
        out = np.zeros((cnt, OBS_DIM))
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_get_obs64(self.h, int(first_env), cnt, _np_ptr(out, C.c_double)))
        return out

    def get_reward64(self, first_env=0, count=None):
        cnt = self.n_envs - first_env if count is None else count
        out = np.zeros(cnt)
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_get_reward64(self.h, int(first_env), cnt, _np_ptr(out, C.c_double)))
        return out

    def enable_trajectory(self, max_substeps=None):
        """Record the per-sub-step positions of every step (robot.trajectory, marinenav_env.py:211-212); f64 handles."""
        self._check(self.L.mn_enable_trajectory(self.h, int(self.params.N if max_substeps is None else max_substeps)))

    def get_trajectory(self, first_env=0, count=None):
        """[count, N, 2] positions after each of the N sub-steps of the last step()."""
        cnt = self.n_envs - first_env if count is None else count
        out = np.zeros((cnt, int(self.params.N), 2))
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_get_trajectory(self.h, int(first_env), cnt, int(self.params.N), _np_ptr(out, C.c_double)))
        return out

    def peek_next_double(self, first_env=0, count=None):
        cnt = self.n_envs - first_env if count is None else count
        out = np.zeros(cnt)
        torch.cuda.synchronize(self.device)
        self._check(self.L.mn_peek_next_double(self.h, int(first_env), cnt, _np_ptr(out, C.c_double)))
        return out

    # ---- profiling hook ----------------------------------------------------------------------
    def profile_begin(self, max_launches):
        self._check(self.L.mn_profile_begin(self.h, int(max_launches)))

    def profile_end(self):
        ms = C.c_double(); n = C.c_int32()
        self._check(self.L.mn_profile_end(self.h, self._stream(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    def profile_reset_end(self):
        """Mean duration / count of the mn_reset_done launches recorded since profile_begin (call before profile_end)."""
        ms = C.c_double(); n = C.c_int32()
        self._check(self.L.mn_profile_reset_end(self.h, self._stream(), C.byref(ms), C.byref(n)))
        return ms.value, n.value

    @staticmethod
    def info_string(code):
        return INFO_STRINGS[int(code)]
