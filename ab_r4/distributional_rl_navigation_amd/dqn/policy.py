"""Inference-side counterpart of the reference's DQN baseline (run_experiments.py:74-98, 367-376).

The reference's DQN agent is its modified stable-baselines3 `ObsEncoderPolicy`: the features extractor is the
whole 26 -> (16 | 16 | 176) -> 64 -> 64 -> 9 observation network WITHOUT activations on the three encoders
(thirdparty/stable_baselines3/common/torch_layers.py:96-135), followed by sb3's default 9 -> 64 -> 64 -> 9 Q head
(dqn/policies.py:48-58, torch_layers.py:137-174).  Only the greedy policy is needed on the batched path (the
experiment sweep and evaluation); DQN training stays with sb3.

Module / parameter names mirror sb3's (`q_net.features_extractor.*`, `q_net.q_net.{0,2,4}.*`) so that the
`policy.pth` inside an sb3 checkpoint zip loads unchanged.
"""
import io
import zipfile

import numpy as np
import torch
import torch.nn as nn


class _Extractor(nn.Module):
    def __init__(self, state_size=26, action_size=9):
        super().__init__()
        assert state_size == 26, "observation dimension needs to be 26 (velocity, goal, measurements)"
        self.velocity_encoder = nn.Linear(2, 16)
        self.goal_encoder = nn.Linear(2, 16)
        self.sensor_encoder = nn.Linear(22, 176)
        self.hidden_layer = nn.Linear(208, 64)
        self.hidden_layer_2 = nn.Linear(64, 64)
        self.output_layer = nn.Linear(64, action_size)

    def forward(self, x):
        f = torch.cat((self.velocity_encoder(x[:, :2]), self.goal_encoder(x[:, 2:4]), self.sensor_encoder(x[:, 4:])), 1)
        return self.output_layer(torch.relu(self.hidden_layer_2(torch.relu(self.hidden_layer(f)))))


class _QNet(nn.Module):
    def __init__(self, state_size, action_size, net_arch):
        super().__init__()
        self.features_extractor = _Extractor(state_size, action_size)
        layers, d = [], action_size
        for h in net_arch:
            layers += [nn.Linear(d, h), nn.ReLU()]
            d = h
        layers.append(nn.Linear(d, action_size))
        self.q_net = nn.Sequential(*layers)

    def forward(self, x):
        return self.q_net(self.features_extractor(x))


class DQNPolicy(nn.Module):
    """Greedy DQN policy over device-resident observation batches."""

    def __init__(self, state_size=26, action_size=9, net_arch=(64, 64), device="cuda:0"):
        super().__init__()
        self.state_size, self.action_size = state_size, action_size
        self.q_net = _QNet(state_size, action_size, list(net_arch))
        self.device = torch.device(device)
        self.to(self.device)
        self.eval()

    @classmethod
    def load(cls, path, device="cuda:0"):
        """`path`: an sb3 checkpoint .zip (its policy.pth is read), a bare policy.pth, or an .npz of the q_net.*
        tensors (tests/golden/pretrained_DQN_seed3/q_net.npz)."""
        if path.endswith(".npz"):
            sd = {k: torch.from_numpy(v) for k, v in np.load(path).items()}
        elif zipfile.is_zipfile(path) and "policy.pth" in zipfile.ZipFile(path).namelist():
            with zipfile.ZipFile(path) as z:
                sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu")
        else:
            sd = torch.load(path, map_location="cpu")
        sd = {k: v for k, v in sd.items() if k.startswith("q_net.")}      # q_net_target.* is training state
        pol = cls(device=device)
        pol.load_state_dict(sd, strict=True)
        return pol

    use_fused_act = True      # GPU tensors: the whole network + argmax as ONE HIP launch (csrc/dqn_act.hip); False = eager PyTorch

    def _fused(self, obs, want_q, want_a):
        """C-ABI mn_dqn_act on a contiguous float32 device batch; the permuted weight image is rebuilt when a parameter was written
        (PyTorch version counters) or re-allocated."""
        import ctypes as C
        from .. import _capi
        L = _capi.lib()
        ex, qn = self.q_net.features_extractor, self.q_net.q_net
        mods = (ex.velocity_encoder, ex.goal_encoder, ex.sensor_encoder, ex.hidden_layer, ex.hidden_layer_2, ex.output_layer, qn[0], qn[2], qn[4])
        ps = [t for m in mods for t in (m.weight, m.bias)]
        sig = tuple((t.data_ptr(), t._version) for t in ps)
        st = getattr(self, "_fused_state", None)
        if st is None or st["image"].device != obs.device:
            st = dict(image=torch.empty(L.mn_dqn_image_floats(), dtype=torch.float32, device=obs.device), sig=None, ptrs=(C.c_void_p * 18)())
            object.__setattr__(self, "_fused_state", st)
        repack = sig != st["sig"]
        if repack:
            for i, t in enumerate(ps):
                assert t.is_cuda and t.is_contiguous() and t.dtype == torch.float32
                st["ptrs"][i] = t.data_ptr()
            st["sig"] = sig
        n = obs.shape[0]
        q = torch.empty(n, self.action_size, dtype=torch.float32, device=obs.device) if want_q else None
        a = torch.empty(n, dtype=torch.int32, device=obs.device) if want_a else None
        p = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
        rc = L.mn_dqn_act(p(obs), st["ptrs"], p(st["image"]), int(repack), p(q), p(a), n, C.c_void_p(torch.cuda.current_stream(obs.device).cuda_stream))
        if rc:
            raise _capi.MarineNavHipError(f"mn_dqn_act failed ({rc})")
        return q, a

    def _fusable(self, obs):
        return (self.use_fused_act and obs.is_cuda and not torch.is_grad_enabled() and len(self.q_net.q_net) == 5
                and self.q_net.q_net[0].out_features == 64 and self.q_net.q_net[2].out_features == 64 and obs.shape[0] > 0)

    @torch.no_grad()
    def q_values(self, obs):
        obs = obs.to(self.device, torch.float32).view(-1, self.state_size)
        if self._fusable(obs):
            return self._fused(obs.contiguous(), True, False)[0]
        return self.q_net(obs)

    @torch.no_grad()
    def act_batch(self, obs):
        """QNetwork._predict (dqn/policies.py:69-73): argmax_a Q(obs, a), one int32 per row."""
        obs = obs.to(self.device, torch.float32).view(-1, self.state_size)
        if self._fusable(obs):
            return self._fused(obs.contiguous(), False, True)[1]
        return self.q_net(obs).argmax(dim=1).to(torch.int32)

    exploration_rate = 0.05      # sb3 DQN's value after its exploration schedule (exploration_final_eps, dqn/dqn.py:82)

    def predict(self, observation, deterministic=True):
        """sb3 surface used by run_experiments.py:86: `action, _ = agent.predict(obs, deterministic=True)`.  With
        deterministic=False, DQN.predict's epsilon-greedy (dqn/dqn.py:249-257): ONE `np.random.rand()` draw per call decides whether
        the whole (vector of) observation(s) gets uniformly random actions instead of the greedy ones."""
        obs = torch.as_tensor(np.asarray(observation), dtype=torch.float32)
        single = obs.dim() == 1
        if not deterministic and np.random.rand() < self.exploration_rate:
            n = 1 if single else obs.view(-1, self.state_size).shape[0]
            a = np.array([np.random.randint(self.action_size) for _ in range(n)], dtype=np.int64)
            return (a[0] if single else a), None
        a = self.act_batch(obs.view(-1, self.state_size)).cpu().numpy().astype(np.int64)
        return (a[0] if single else a), None
