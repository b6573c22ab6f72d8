"""Golden vectors for the DQN baseline policy.  Runs in the BUILD container only.

Imports the reference's modified stable-baselines3 (`/root/reference/thirdparty/stable_baselines3`) with a
permissive stub `gym` package (gym is not installed; only class names are needed at import time), builds the
reference's own Q-network (dqn/policies.py:19-70 `QNetwork` pieces: torch_layers.py:96-135 `ObsEncoder` +
torch_layers.py:137-174 `create_mlp(9, 9, [64, 64])`), loads the weights of the shipped DQN checkpoint
(`pretrained_models/DQN/seed_3/latest_model.zip` -> policy.pth) and records

  g10_dqn.npz         obs [512,26] f32 (the g3 single-step observations), q [512,9] f32, action [512] int64
                      eval_actions [30,Lmax] int8 (-1 padded), eval_len [30], eval_rewards [30], eval_successes [30]:
                      the LAST row of the checkpoint's own `evaluations.npz` (the greedy episodes on the 30
                      seed-348 evaluation worlds = tests/golden/eval_config_seed3.json, recorded right before
                      `latest_model.zip` was written: callbacks.py:500-543).  NOTE the authors' forward pass ran
                      on their GPU (TF32-era torch): recorded actions deviate from an exact-f32 forward wherever
                      the top-2 Q gap is below ~0.03, so closed-loop tests accept a first divergence only there.
  pretrained_DQN_seed3/q_net.npz   the 18 q_net.* tensors of the checkpoint (data, ~100 KB)
  g11_dqn_train.npz   one gradient step of the reference's own `DQN.train` (dqn/dqn.py:188-230) on its own
                      `ObsEncoderPolicy` (dqn/policies.py:212-240; Adam lr 1e-4, smooth-L1 loss, clip_grad_norm_ 10):
                      q_net = the checkpoint, target = checkpoint + seeded noise, a fixed batch of 32 transitions ->
                      loss, clipped q_net gradients, q_net parameters after the step.  The DQN object is assembled
                      around the real policy without an env (only the attributes `train` reads), so the code that runs
                      is the reference's.

    python tests/golden/make_golden_dqn.py
"""
import importlib
import importlib.abc
import importlib.machinery
import io
import os
import sys
import types
import zipfile

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


class _Space:                      # stable stand-ins for gym.spaces.* (isinstance checks inside sb3)
    def __init__(self, shape=None, dtype=None):
        self.shape, self.dtype = shape, dtype


class _Box(_Space):
    def __init__(self, low, high, shape=None, dtype=np.float32):
        super().__init__(shape, dtype)
        self.low, self.high = low, high


class _Discrete(_Space):
    def __init__(self, n):
        super().__init__((), np.int64)
        self.n = n


_STABLE = {"gym.spaces": dict(Space=_Space, Box=_Box, Discrete=_Discrete, MultiDiscrete=type("MultiDiscrete", (_Space,), {}),
                              MultiBinary=type("MultiBinary", (_Space,), {}), Dict=type("Dict", (_Space,), {}),
                              Tuple=type("Tuple", (_Space,), {}))}


def _stub_module(name):
    m = types.ModuleType(name)

    def ga(attr, _n=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        if _n in _STABLE and attr in _STABLE[_n]:
            return _STABLE[_n][attr]
        if attr[0].islower():
            return importlib.import_module(_n + "." + attr)
        cls = type(attr, (), {})
        setattr(m, attr, cls)
        return cls

    m.__getattr__ = ga
    m.__path__ = []
    return m


class _GymFinder(importlib.abc.MetaPathFinder, importlib.abc.Loader):
    def find_spec(self, name, path, target=None):
        if name == "gym" or name.startswith("gym."):
            return importlib.machinery.ModuleSpec(name, self, is_package=True)

    def create_module(self, spec):
        return _stub_module(spec.name)

    def exec_module(self, m):
        pass


def main():
    sys.meta_path.insert(0, _GymFinder())
    import gym
    gym.__version__ = "0.21.0"
    sys.path.insert(0, os.path.join(REF, "thirdparty"))
    from stable_baselines3.common.torch_layers import ObsEncoder, create_mlp

    with zipfile.ZipFile(os.path.join(REF, "pretrained_models/DQN/seed_3/latest_model.zip")) as z:
        sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu")
    extractor = ObsEncoder(observation_space=None, state_size=26, action_size=9)
    head = torch.nn.Sequential(*create_mlp(9, 9, [64, 64]))
    extractor.load_state_dict({k[len("q_net.features_extractor."):]: v for k, v in sd.items()
                               if k.startswith("q_net.features_extractor.")})
    head.load_state_dict({k[len("q_net.q_net."):]: v for k, v in sd.items() if k.startswith("q_net.q_net.")})

    g3 = np.load(os.path.join(OUT, "g3_single_step.npz"))
    obs = np.ascontiguousarray(g3["obs"][:512].astype(np.float32))
    with torch.no_grad():
        q = head(extractor(torch.from_numpy(obs)))          # QNetwork.forward (policies.py:60-67)
        action = q.argmax(dim=1).reshape(-1)                # QNetwork._predict (policies.py:69-73)
    ev = np.load(os.path.join(REF, "pretrained_models/DQN/seed_3/evaluations.npz"), allow_pickle=True)
    rec = [np.asarray(a, dtype=np.int64) for a in ev["actions"][-1]]
    lens = np.array([len(a) for a in rec], dtype=np.int32)
    acts = np.full((len(rec), int(lens.max())), -1, dtype=np.int8)
    for i, a in enumerate(rec):
        acts[i, :len(a)] = a
    np.savez_compressed(os.path.join(OUT, "g10_dqn.npz"), obs=obs, q=q.numpy(), action=action.numpy(),
                        eval_actions=acts, eval_len=lens, eval_rewards=ev["rewards"][-1].astype(np.float64),
                        eval_successes=ev["successes"][-1].astype(np.uint8))
    os.makedirs(os.path.join(OUT, "pretrained_DQN_seed3"), exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "pretrained_DQN_seed3", "q_net.npz"),
                        **{k: v.numpy() for k, v in sd.items() if k.startswith("q_net.")})
    print("g10_dqn:", obs.shape, q.shape, "actions hist", np.bincount(action.numpy(), minlength=9))

    # ---- G11: one step of the reference's DQN.train --------------------------------------------------------------
    from collections import namedtuple
    from stable_baselines3 import DQN
    from stable_baselines3.dqn.policies import ObsEncoderPolicy
    torch.manual_seed(0)
    pol = ObsEncoderPolicy(_Box(-np.inf, np.inf, shape=(26,), dtype=np.float32), _Discrete(9), lr_schedule=lambda _: 1e-4)
    pol.load_state_dict(sd)
    gen = torch.Generator(); gen.manual_seed(5)
    with torch.no_grad():
        for p_ in pol.q_net_target.parameters():
            p_.add_(0.02 * torch.randn(p_.shape, generator=gen))
    B = 32
    rs = np.random.RandomState(11)
    batch = dict(observations=torch.from_numpy(np.ascontiguousarray(g3["obs"][100:100 + B].astype(np.float32))),
                 next_observations=torch.from_numpy(np.ascontiguousarray(g3["obs"][200:200 + B].astype(np.float32))),
                 actions=torch.from_numpy(rs.randint(0, 9, size=(B, 1)).astype(np.int64)),
                 rewards=torch.from_numpy((rs.randn(B, 1) * 2 - 1).astype(np.float32)),
                 dones=torch.from_numpy((rs.rand(B, 1) < 0.2).astype(np.float32)))
    Samples = namedtuple("Samples", ["observations", "actions", "next_observations", "dones", "rewards"])

    class _Buffer:
        def sample(self, batch_size, env=None):
            assert batch_size == B
            return Samples(**batch)

    class _Logger:
        def record(self, *a, **k):
            pass

    m = DQN.__new__(DQN)
    m.policy, m.q_net, m.q_net_target = pol, pol.q_net, pol.q_net_target
    m.gamma, m.max_grad_norm, m._n_updates = 0.99, 10, 0
    m.lr_schedule, m._current_progress_remaining, m._vec_normalize_env = (lambda _: 1e-4), 1.0, None
    m._logger, m.replay_buffer = _Logger(), _Buffer()
    target_before = {k: v.clone() for k, v in pol.q_net_target.state_dict().items()}
    m.train(gradient_steps=1, batch_size=B)          # the reference's own code (dqn/dqn.py:188-230)
    out = {f"batch_{k}": v.numpy() for k, v in batch.items()}
    out.update({f"tgt_{k}": v.numpy() for k, v in target_before.items()})
    out.update({f"grad_{k}": p_.grad.numpy() for k, p_ in pol.q_net.named_parameters()})
    out.update({f"after_{k}": p_.detach().numpy() for k, p_ in pol.q_net.named_parameters()})
    # loss of the step: re-evaluate with the checkpoint weights through the reference network (same ops as dqn.py:199-216)
    ref = ObsEncoderPolicy(_Box(-np.inf, np.inf, shape=(26,), dtype=np.float32), _Discrete(9), lr_schedule=lambda _: 1e-4)
    ref.load_state_dict(sd)
    ref.q_net_target.load_state_dict(target_before)
    with torch.no_grad():
        nq = ref.q_net_target(batch["next_observations"]).max(dim=1)[0].reshape(-1, 1)
        tq = batch["rewards"] + (1 - batch["dones"]) * 0.99 * nq
        cq = torch.gather(ref.q_net(batch["observations"]), dim=1, index=batch["actions"])
        out["loss"] = np.float32(torch.nn.functional.smooth_l1_loss(cq, tq).item())
    np.savez_compressed(os.path.join(OUT, "g11_dqn_train.npz"), **out)
    print("g11_dqn_train: loss", out["loss"], "n_updates", m._n_updates,
          "max|grad|", max(float(np.abs(v).max()) for k, v in out.items() if k.startswith("grad_")))


if __name__ == "__main__":
    main()
