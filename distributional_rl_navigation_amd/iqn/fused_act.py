"""Python side of the fused IQN action-value kernel (csrc/iqn_act.hip, `mn_iqn_qvals`)."""
import ctypes as C

import torch

from .. import _capi


def _p(t):
    return C.c_void_p(t.data_ptr())


@torch.no_grad()
def fused_qvals(net, states, cvar=1.0, taus=None, generator=None):
    """Q(s, .) = mean over K = 32 quantile samples (model.py:188-191) for states [n, 26] on the GPU.

    The three observation encoders (26 -> 208, model.py:170-173) run in PyTorch; the cosine
    embedding, Hadamard product, the three hidden layers and the mean over taus run in one HIP kernel.
    `taus` [n, 32] may be injected (tests); otherwise they are drawn on the device generator.
    """
    assert states.is_cuda and states.dtype == torch.float32
    n = states.shape[0]
    K = net.K
    feats = torch.cat((net.velocity_encoder(states[:, :2]), net.goal_encoder(states[:, 2:4]),
                       net.sensor_encoder(states[:, 4:])), 1).contiguous()
    if taus is None:
        taus = torch.rand(n, K, device=states.device, generator=generator)
    taus = taus.to(device=states.device, dtype=torch.float32)
    if torch.is_tensor(cvar):
        taus = taus * cvar.to(states.device).view(-1, 1)
    elif cvar != 1.0:
        taus = taus * cvar
    taus = taus.contiguous()
    q = torch.empty(n, net.action_size, dtype=torch.float32, device=states.device)
    w = [net.cos_embedding.weight, net.cos_embedding.bias, net.hidden_layer.weight, net.hidden_layer.bias,
         net.hidden_layer_2.weight, net.hidden_layer_2.bias, net.output_layer.weight, net.output_layer.bias]
    for t in w:
        assert t.is_contiguous() and t.dtype == torch.float32 and t.is_cuda
    stream = C.c_void_p(torch.cuda.current_stream(states.device).cuda_stream)
    rc = _capi.lib().mn_iqn_qvals(_p(feats), _p(taus), *[_p(t) for t in w], _p(q), n, K, stream)
    if rc:
        raise _capi.MarineNavHipError(f"mn_iqn_qvals failed ({rc})")
    return q
