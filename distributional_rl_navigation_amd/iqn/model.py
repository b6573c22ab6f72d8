"""IQN network (`ObsEncoder`) in PyTorch-ROCm.

Same architecture, parameter names and two-file checkpoint format as the reference's
thirdparty/IQN/model.py:111-225, so reference checkpoints (network_params.pth +
constructor_params.json) load unchanged.  The small dense layers run on MFMA through
hipBLASLt; no custom kernels here (north-star keeps the learner in PyTorch-ROCm).
"""
import json
import math
import os

import torch
import torch.nn as nn


class ObsEncoder(nn.Module):
    """Encoders for (velocity, goal, sonar) + cosine quantile embedding (model.py:111-186)."""

    def __init__(self, state_size, action_size, seed, device="cpu"):
        super().__init__()
        self.device = torch.device(device)
        self.seed_id = seed
        self.seed = torch.manual_seed(seed)  # model.py:117 (makes local and target nets identical)
        self.K = 32                          # model.py:118

        self.state_size = state_size
        self.action_size = action_size
        if state_size != 26:
            raise ValueError(f"ObsEncoder is built for the 26-value marinenav observation (2 velocity + 2 goal + 11 x 2 sonar), got state_size = {state_size}")
        self.velocity_encoder = nn.Linear(2, 16)
        self.goal_encoder = nn.Linear(2, 16)
        self.sensor_encoder = nn.Linear(22, 176)

        self.register_buffer("pis", torch.tensor([math.pi * i for i in range(64)], dtype=torch.float32).view(1, 1, 64),
                             persistent=False)
        self.cos_embedding = nn.Linear(64, 208)

        self.hidden_layer = nn.Linear(208, 64)
        self.hidden_layer_2 = nn.Linear(64, 64)
        self.output_layer = nn.Linear(64, action_size)
        self.to(self.device)

    def calc_cos(self, batch_size, n_tau=8, cvar=1.0, taus=None):
        """model.py:141-157.  `taus` [B, n_tau] may be injected (tests); otherwise they are drawn on
        the DEVICE generator (the reference draws on the CPU generator and copies, model.py:149).
        `cvar` is a float or a per-row tensor [B] (batched adaptive CVaR)."""
        dev = self.pis.device
        if taus is None:
            taus = torch.rand(batch_size, n_tau, device=dev)
        taus = taus.to(dev).unsqueeze(-1)
        if torch.is_tensor(cvar):
            taus = taus * cvar.to(dev).view(-1, 1, 1)
        elif cvar != 1.0:                      # x * 1.0 == x: skip the kernel
            taus = taus * cvar
        cos = torch.cos(taus * self.pis)
        return cos, taus

    def forward(self, inputs, num_tau=8, cvar=1.0, taus=None):
        """model.py:160-186 -> (quantiles [B, num_tau, A], taus [B, num_tau, 1])."""
        if inputs.shape[1] != self.state_size:
            raise ValueError(f"ObsEncoder.forward: rows of {inputs.shape[1]} values, the network takes {self.state_size}")
        batch_size = inputs.shape[0]
        v_features = self.velocity_encoder(inputs[:, :2])
        g_features = self.goal_encoder(inputs[:, 2:4])
        s_features = self.sensor_encoder(inputs[:, 4:])
        features = torch.cat((v_features, g_features, s_features), 1)   # no activation (App. A A3)

        cos, taus = self.calc_cos(batch_size, num_tau, cvar, taus)
        cos_features = torch.relu(self.cos_embedding(cos.view(batch_size * num_tau, 64))).view(batch_size, num_tau, 208)
        x = (features.unsqueeze(1) * cos_features).view(batch_size * num_tau, 208)
        x = torch.relu(self.hidden_layer(x))
        x = torch.relu(self.hidden_layer_2(x))
        out = self.output_layer(x)
        return out.view(batch_size, num_tau, self.action_size), taus

    def get_qvals(self, inputs, cvar=1.0, taus=None):
        quantiles, _ = self.forward(inputs=inputs, num_tau=self.K, cvar=cvar, taus=taus)  # model.py:188-191
        return quantiles.mean(dim=1)

    def get_constructor_parameters(self):
        return dict(state_size=self.state_size, action_size=self.action_size, seed=self.seed_id)

    def save(self, directory, prefix=""):
        """The reference's two-file checkpoint (model.py:198-207): `network_params.pth` (state dict) + `constructor_params.json`; `prefix` names a second
        pair beside it ("best_": the best evaluation so far, `IQNAgent.learn_vec`)."""
        # clone: the parameters may be views of one flat buffer (iqn/fused_train.py); upstream checkpoints hold one
        # independent storage per tensor
        torch.save({k: v.detach().clone() for k, v in self.state_dict().items()}, os.path.join(directory, prefix + "network_params.pth"))
        with open(os.path.join(directory, prefix + "constructor_params.json"), mode="w") as f:
            json.dump(self.get_constructor_parameters(), f)

    @classmethod
    def load(cls, directory, device="cpu", prefix=""):
        """Rebuild a network from a checkpoint directory written by `save` or by the reference (model.py:209-225): the constructor arguments from the JSON
        file, then the weights."""
        with open(os.path.join(directory, prefix + "constructor_params.json")) as f:
            kwargs = dict(json.load(f), device=device)
        net = cls(**kwargs)
        net.load_state_dict(torch.load(os.path.join(directory, prefix + "network_params.pth"), map_location=device))
        return net.to(device)
