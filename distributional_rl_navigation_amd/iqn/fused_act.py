"""Python side of the fused IQN act kernel (csrc/iqn_act.hip, `mn_iqn_act` / `mn_iqn_act_rng`).

Every network that acts through the kernel owns one `ActContext` (C-ABI `mn_iqn_ctx`): the permuted weight image
the kernel stages into LDS is cached there and rebuilt only after the weights changed.  Changes made through
PyTorch (optimizer.step, load_state_dict, copy_) are detected from the parameters' version counters; writers that
bypass PyTorch (the fused HIP Adam step, csrc/iqn_train.hip) call `weights_changed(net)`.
"""
import ctypes as C
import weakref

import torch

from .. import _capi

_ORDER = ("velocity_encoder", "goal_encoder", "sensor_encoder", "cos_embedding", "hidden_layer", "hidden_layer_2", "output_layer")


def _p(t):
    return C.c_void_p(t.data_ptr()) if t is not None else None


def _params(net):
    # (through the module / parameter dicts: nn.Module.__getattr__ is a Python-level fallback lookup, 28 of them per act call were 15 us of a 90-us vector step at 4 096 envs)
    mods = net._modules
    out = []
    for name in _ORDER:
        ps = mods[name]._parameters
        out.append(ps["weight"]); out.append(ps["bias"])
    return out


class ActContext:
    """`mn_iqn_ctx` of one network on one device."""

    def __init__(self, device):
        self.device = torch.device(device)
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = _capi.lib().mn_iqn_create(C.byref(h))
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_create failed ({rc})")
        self.h = h
        self._sig = None
        self.variant = self.DEFAULT_VARIANT
        self.tau_mode = 0
        self._ptrs = (C.c_void_p * 14)()
        self._fin = weakref.finalize(self, _capi.lib().mn_iqn_destroy, h)

    def __deepcopy__(self, memo):      # a copied network gets its own context lazily (act_context)
        return None

    def __reduce__(self):
        return (type(None), ())

    def weights(self, net):
        """HOST array of the 14 device pointers; marks the cached image stale when a parameter was re-allocated or
        written through PyTorch since the last call."""
        ps = _params(net)
        sig = tuple((t.data_ptr(), t._version) for t in ps)
        if sig != self._sig:
            for i, t in enumerate(ps):
                assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
                self._ptrs[i] = t.data_ptr()
            self._sig = sig
            self.invalidate()
        return self._ptrs

    def invalidate(self):
        _capi.lib().mn_iqn_weights_changed(self.h)

    DEFAULT_VARIANT = 2

    def set_variant(self, variant):
        """2 = the split-f16 kernel (default: three f16 MFMA products per float32 product, float32-class accuracy, ~3x faster),
        0 = the exact-f32 16x16x4 MFMA kernel (A / B measurements, tests)."""
        rc = _capi.lib().mn_iqn_set_variant(self.h, int(variant))
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_set_variant failed ({rc})")
        self.variant = int(variant)

    def set_tau_mode(self, mode):
        """0 = every observation row its own 32 taus (default, the reference's per-call draw); 1 = one set of 32 taus per launch:
        layer 1 of the network becomes a constant of the launch (C-ABI mn_iqn_set_tau_mode; split-f16 kernel only); 2 = the same
        with the wavefront-per-row kernel for every batch size (1 switches to the environment-tiled kernel for large batches); 3 = the
        environment-tiled kernel for every batch size."""
        if int(mode) == self.tau_mode:
            return
        rc = _capi.lib().mn_iqn_set_tau_mode(self.h, int(mode))
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_set_tau_mode failed ({rc})")
        self.tau_mode = int(mode)

    def set_grid(self, max_workgroups):
        """0 = one persistent workgroup per CU (default); > 0 = up to that many shorter workgroups, so that other streams'
        kernels get CUs while an act launch is in flight (C-ABI mn_iqn_set_grid)."""
        rc = _capi.lib().mn_iqn_set_grid(self.h, int(max_workgroups))
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_set_grid failed ({rc})")
        self.__dict__.pop("_late_ok", None)      # (late_rows_possible: rows per wavefront follow the grid)

    def refresh(self, net):
        """Rebuild the cached weight image now (current stream) if it is stale -- see mn_iqn_refresh."""
        stream = _capi.stream_ptr(self.device)
        rc = _capi.lib().mn_iqn_refresh(self.h, self.weights(net), stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_refresh failed ({rc})")

    def profile_begin(self, max_launches):
        rc = _capi.lib().mn_iqn_profile_begin(self.h, int(max_launches))
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_profile_begin failed ({rc})")

    def profile_end(self):
        ms, nl = C.c_double(), C.c_int32()
        stream = _capi.stream_ptr(self.device)
        rc = _capi.lib().mn_iqn_profile_end(self.h, stream, C.byref(ms), C.byref(nl))
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_profile_end failed ({rc})")
        return ms.value, nl.value


def act_context(net):
    """The network's own act context (created on first use, on the device its parameters live on)."""
    dev = net.output_layer.weight.device
    ctx = getattr(net, "_act_ctx", None)
    if ctx is None or ctx.device != dev:
        ctx = ActContext(dev)
        object.__setattr__(net, "_act_ctx", ctx)      # not a Module / Parameter: keep it out of state_dict
    return ctx


def weights_changed(net):
    """Tell the act path that `net`'s weights were written outside PyTorch's version tracking (HIP kernels)."""
    ctx = getattr(net, "_act_ctx", None)
    if ctx is not None:
        ctx.invalidate()


def _taus(net, n, device, cvar, taus, generator):
    if taus is None:
        taus = torch.rand(n, net.K, device=device, generator=generator)
    taus = taus.to(device=device, dtype=torch.float32)
    if torch.is_tensor(cvar):
        taus = taus * cvar.to(device).view(-1, 1)
    elif cvar != 1.0:
        taus = taus * cvar
    return taus.contiguous()


class ActRng:
    """State of the library's own tau / exploration draws for `fused_act(..., rng=...)`: {seed, call counter} on the
    device plus the [33 n] scratch buffer the draws of a call are written to."""

    def __init__(self, seed, device):
        self.state = torch.tensor([int(seed) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64, device=device)
        self._draws = {}

    def draws(self, n, K):
        buf = self._draws.get(n)
        if buf is None:
            buf = self._draws[n] = torch.empty(n * (K + 1), dtype=torch.float32, device=self.state.device)
        return buf


def _arm_late_rows(ctx, late_env, n, want_quantiles):
    """`late_env`: a VecMarineNavEnv whose `reset_done(under_next_act=True)` is still running on its own stream while `states` (its
    observation buffer) goes into this act launch.  The launch is told which rows are being rewritten (mn_iqn_set_late_rows) and takes them
    last; a kernel form that cannot do that waits for the reset instead.  Returns True if the launch that follows must be joined afterwards."""
    if late_env is None:
        return False
    lr = late_env.take_late_rows()
    if lr is None:
        return False
    mask, ready, tick = lr
    rc = 1 if want_quantiles else _capi.lib().mn_iqn_set_late_rows(ctx.h, _p(mask), C.c_void_p(ready), C.c_uint32(tick), n)
    if rc == 1:
        late_env.join_reset()
        return False
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_set_late_rows failed ({rc})")
    return True


def late_rows_possible(net, n, shared_taus=False):
    """Whether an act launch of `n` rows of `net` takes late rows (the split-f16 kernel with per-row taus, at most 64 rows per wavefront): what
    `IQNAgent.vec_step` asks before it puts an episode reset under the next act kernel -- for any other form the reset stays in front."""
    if shared_taus:
        return False
    ctx = act_context(net)
    key = (int(n), ctx.variant)
    cache = ctx.__dict__.setdefault("_late_ok", {})
    if key not in cache:
        probe = torch.zeros(1, dtype=torch.int32, device=ctx.device)      # (pointers are not dereferenced by the query)
        mode, ctx.tau_mode = ctx.tau_mode, None
        ctx.set_tau_mode(0)
        rc = _capi.lib().mn_iqn_set_late_rows(ctx.h, _p(probe), _p(probe), C.c_uint32(0), int(n))
        _capi.lib().mn_iqn_set_late_rows(ctx.h, None, None, C.c_uint32(0), 0)
        if mode is not None:
            ctx.set_tau_mode(mode)
        cache[key] = rc == 0
    return cache[key]


def late_timeouts(net):
    """Waits of late rows that ran out in `net`'s act context (0 in a healthy run): C-ABI mn_iqn_late_timeouts.  Synchronises the stream."""
    ctx = act_context(net)
    out = C.c_uint32()
    dev = next(net.parameters()).device
    rc = _capi.lib().mn_iqn_late_timeouts(ctx.h, _capi.stream_ptr(dev), C.byref(out))
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_late_timeouts failed ({rc})")
    return out.value


def late_timeouts_peek(net):
    """The same count WITHOUT synchronising (mn_iqn_late_timeouts_peek: what the launches executed so far have reported through a host-mapped word)."""
    out = C.c_uint32()
    rc = _capi.lib().mn_iqn_late_timeouts_peek(act_context(net).h, C.byref(out))
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_late_timeouts_peek failed ({rc})")
    return out.value


def set_late_bound_ms(net, ms):
    """Bound of a late row's wait for its reset (mn_iqn_set_late_bound_ms; default 500 ms)."""
    rc = _capi.lib().mn_iqn_set_late_bound_ms(act_context(net).h, C.c_double(ms))
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_set_late_bound_ms failed ({rc})")


@torch.no_grad()
def fused_act(net, states, eps=0.0, cvar=1.0, taus=None, generator=None, want_qvals=False, rng=None, want_quantiles=False,
              shared_taus=False, late_env=None):
    """IQNAgent.act for states [n, 26] on the GPU in ONE kernel: encoders, cosine embedding, Hadamard
    product, hidden layers, mean over K = 32 taus, argmax and epsilon-greedy.
    Returns actions [n] int32; with want_qvals (actions, Q [n, 9]); with want_quantiles -- the batched
    IQNAgent.act_eval (agent.py:217-236) -- (actions, quantiles [n, 32, 9], taus [n, 32, 1]) (+ Q if want_qvals).
    `taus` [n, 32] may be injected; otherwise, with `rng` (an ActRng) the library draws taus and exploration uniforms
    in its own preparation launch (no torch.rand kernels), and without it they come from torch.rand on `generator`.
    `shared_taus` (opt-in): ONE set of 32 taus (x the scalar `cvar`) for all n rows of the call instead of 32 per row -- layer 1 of
    the network becomes a constant of the launch (mn_iqn_set_tau_mode; 216 instead of 372 matrix instructions per row).  Injected
    `taus` are then [32]; a per-row `cvar` tensor (adaptive policies) needs per-row taus and keeps the default mode."""
    assert states.is_cuda and states.dtype == torch.float32 and states.is_contiguous()
    n = states.shape[0]
    dev = states.device
    ctx = act_context(net)
    shared = bool(shared_taus) and not torch.is_tensor(cvar) and ctx.variant == 2
    # mode 1: the library picks the kernel form (from 65 536 rows up the MFMA columns are environments, iqn_act_tiled.h; below, one
    # wavefront per row); shared_taus="wave" / "tiled" pin a form (modes 2 / 3: A / B measurements, tests)
    ctx.set_tau_mode({"wave": 2, "tiled": 3}.get(shared_taus, 1) if shared else 0)
    actions = torch.empty(n, dtype=torch.int32, device=dev)
    q = torch.empty(n, net.action_size, dtype=torch.float32, device=dev) if want_qvals else None
    quant = torch.empty(n, net.K, net.action_size, dtype=torch.float32, device=dev) if want_quantiles else None
    stream = _capi.stream_ptr(dev)
    joined_after = _arm_late_rows(ctx, late_env, n, want_quantiles)      # (after set_tau_mode: the form of THIS launch decides)
    if taus is None and rng is not None:
        cv_row = cvar.to(device=dev, dtype=torch.float32).contiguous() if torch.is_tensor(cvar) else None
        draws = rng.draws(n, net.K)
        rc = _capi.lib().mn_iqn_act_rng(ctx.h, _p(states), ctx.weights(net), _p(rng.state), _p(draws), _p(cv_row),
                                        C.c_float(1.0 if cv_row is not None else float(cvar)), C.c_float(float(eps)),
                                        _p(actions), _p(q), _p(quant), n, net.K, stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_act_rng failed ({rc})")
        t = draws[:net.K].view(1, net.K).expand(n, net.K) if shared else draws[:n * net.K].view(n, net.K)
    elif shared:
        t1 = (torch.rand(net.K, device=dev, generator=generator) if taus is None else taus.to(device=dev, dtype=torch.float32).reshape(-1))
        assert t1.numel() == net.K, "shared_taus: one row of K taus"
        t1 = (t1 * cvar if cvar != 1.0 else t1).contiguous()
        u = torch.rand(n, device=dev, generator=generator) if eps > 0.0 else None
        rc = _capi.lib().mn_iqn_act(ctx.h, _p(states), _p(t1), ctx.weights(net), _p(q), _p(u), C.c_float(float(eps)),
                                    _p(actions), _p(quant), n, net.K, stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_act failed ({rc})")
        t = t1.view(1, net.K).expand(n, net.K)
    else:
        if taus is None and eps > 0.0:      # one RNG launch for the n x K taus and the n exploration uniforms
            buf = torch.rand(n * (net.K + 1), device=dev, generator=generator)
            t = _taus(net, n, dev, cvar, buf[:n * net.K].view(n, net.K), None)
            u = buf[n * net.K:]
        else:
            t = _taus(net, n, dev, cvar, taus, generator)
            u = torch.rand(n, device=dev, generator=generator) if eps > 0.0 else None
        rc = _capi.lib().mn_iqn_act(ctx.h, _p(states), _p(t), ctx.weights(net), _p(q), _p(u), C.c_float(float(eps)),
                                    _p(actions), _p(quant), n, net.K, stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_act failed ({rc})")
    if joined_after:      # later work on this stream (the next step reads every row) comes after the reset launch's end
        late_env.join_reset()
    out = (actions,)
    if want_quantiles:
        out += (quant, t.clone().view(n, net.K, 1))       # (.clone(): the library's draw buffer is reused by the next call)
    if want_qvals:
        out += (q,)
    return out if len(out) > 1 else actions


@torch.no_grad()
def fused_qvals(net, states, cvar=1.0, taus=None, generator=None):
    """Q(s, .) = mean over K = 32 quantile samples (model.py:188-191) for states [n, 26] on the GPU."""
    assert states.is_cuda and states.dtype == torch.float32
    states = states.contiguous()
    n = states.shape[0]
    ctx = act_context(net)
    ctx.set_tau_mode(0)
    t = _taus(net, n, states.device, cvar, taus, generator)
    q = torch.empty(n, net.action_size, dtype=torch.float32, device=states.device)
    stream = _capi.stream_ptr(states.device)
    rc = _capi.lib().mn_iqn_act(ctx.h, _p(states), _p(t), ctx.weights(net), _p(q), None, C.c_float(0.0), None, None, n, net.K,
                                stream)
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_act failed ({rc})")
    return q
