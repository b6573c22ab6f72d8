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


def _stub_module(name):
    m = types.ModuleType(name)

    def ga(attr, _n=name):
        if attr.startswith("__"):
            raise AttributeError(attr)
        if attr[0].islower():
            return importlib.import_module(_n + "." + attr)
        return type(attr, (), {})

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


if __name__ == "__main__":
    main()
