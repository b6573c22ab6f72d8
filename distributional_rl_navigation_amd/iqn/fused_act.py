"""Python side of the fused IQN act kernel (csrc/iqn_act.hip, `mn_iqn_act`)."""
import ctypes as C

import torch

from .. import _capi

_ORDER = ("velocity_encoder", "goal_encoder", "sensor_encoder", "cos_embedding", "hidden_layer", "hidden_layer_2", "output_layer")


def _p(t):
    return C.c_void_p(t.data_ptr())


def _weight_ptrs(net):
    ptrs = (C.c_void_p * 14)()
    i = 0
    for name in _ORDER:
        m = getattr(net, name)
        for t in (m.weight, m.bias):
            assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
            ptrs[i] = t.data_ptr()
            i += 1
    return ptrs


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


@torch.no_grad()
def fused_act(net, states, eps=0.0, cvar=1.0, taus=None, generator=None, want_qvals=False, rng=None):
    """IQNAgent.act for states [n, 26] on the GPU in ONE kernel: encoders, cosine embedding, Hadamard
    product, hidden layers, mean over K = 32 taus, argmax and epsilon-greedy.
    Returns actions [n] int32 (and Q-values [n, 9] if want_qvals).  `taus` [n, 32] may be injected; otherwise, with
    `rng` (an ActRng) the library draws taus and exploration uniforms in the launch that prepares the weights (no
    torch.rand kernels; the draws of the last call stay readable in rng.draws(n, 32)), and without it they come from
    torch.rand on `generator`."""
    assert states.is_cuda and states.dtype == torch.float32 and states.is_contiguous()
    n = states.shape[0]
    dev = states.device
    if taus is None and rng is not None:
        actions = torch.empty(n, dtype=torch.int32, device=dev)
        q = torch.empty(n, net.action_size, dtype=torch.float32, device=dev) if want_qvals else None
        cv_row = cvar.to(device=dev, dtype=torch.float32).contiguous() if torch.is_tensor(cvar) else None
        stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
        rc = _capi.lib().mn_iqn_act_rng(_p(states), _weight_ptrs(net), _p(rng.state), _p(rng.draws(n, net.K)),
                                        _p(cv_row) if cv_row is not None else None,
                                        C.c_float(1.0 if cv_row is not None else float(cvar)), C.c_float(float(eps)), _p(actions),
                                        _p(q) if q is not None else None, n, net.K, stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_act_rng failed ({rc})")
        return (actions, q) if want_qvals else actions
    if taus is None and eps > 0.0:      # one RNG launch for the n x K taus and the n exploration uniforms
        buf = torch.rand(n * (net.K + 1), device=dev, generator=generator)
        t = _taus(net, n, dev, cvar, buf[:n * net.K].view(n, net.K), None)
        u = buf[n * net.K:]
    else:
        t = _taus(net, n, dev, cvar, taus, generator)
        u = torch.rand(n, device=dev, generator=generator) if eps > 0.0 else None
    actions = torch.empty(n, dtype=torch.int32, device=dev)
    q = torch.empty(n, net.action_size, dtype=torch.float32, device=dev) if want_qvals else None
    stream = C.c_void_p(torch.cuda.current_stream(dev).cuda_stream)
    rc = _capi.lib().mn_iqn_act(_p(states), _p(t), _weight_ptrs(net), _p(q) if q is not None else None,
                                _p(u) if u is not None else None, C.c_float(float(eps)), _p(actions), n, net.K, stream)
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_act failed ({rc})")
    return (actions, q) if want_qvals else actions


@torch.no_grad()
def fused_qvals(net, states, cvar=1.0, taus=None, generator=None):
    """Q(s, .) = mean over K = 32 quantile samples (model.py:188-191) for states [n, 26] on the GPU."""
    assert states.is_cuda and states.dtype == torch.float32
    states = states.contiguous()
    n = states.shape[0]
    t = _taus(net, n, states.device, cvar, taus, generator)
    q = torch.empty(n, net.action_size, dtype=torch.float32, device=states.device)
    stream = C.c_void_p(torch.cuda.current_stream(states.device).cuda_stream)
    rc = _capi.lib().mn_iqn_act(_p(states), _p(t), _weight_ptrs(net), _p(q), None, C.c_float(0.0), None, n, net.K, stream)
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_act failed ({rc})")
    return q
