"""Batched APF / BA baselines (planners.py) against actions computed by the reference's APF.py / BA.py
(tests/golden/make_golden.py g9): 2304 observations incl. zero-velocity, no-return, one/two-return and
vertical-wall cases.  Integer outputs: exact match required."""
import os

import numpy as np
import pytest
import torch

from distributional_rl_navigation_amd.planners import apf_act_batch, ba_act_batch

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g9_planners.npz"))
DEVICES = ["cpu", pytest.param("cuda:0", marks=pytest.mark.gpu)]      # the same exact-parity check runs on the device under -m gpu


def _dev(device):
    if device != "cpu" and not torch.cuda.is_available():
        pytest.skip("no GPU")
    return device


@pytest.mark.parametrize("device", DEVICES)
def test_apf_matches_reference(device):
    obs = torch.from_numpy(Z["obs"]).to(_dev(device))
    act = apf_act_batch(obs, Z["a"], Z["w"]).cpu().numpy()
    bad = np.nonzero(act != Z["apf"])[0]
    assert len(bad) == 0, (bad[:10], act[bad[:10]], Z["apf"][bad[:10]])


@pytest.mark.parametrize("device", DEVICES)
def test_ba_matches_reference(device):
    obs = torch.from_numpy(Z["obs"]).to(_dev(device))
    act = ba_act_batch(obs, Z["a"], Z["w"]).cpu().numpy()
    bad = np.nonzero(act != Z["ba"])[0]
    assert len(bad) == 0, (bad[:10], act[bad[:10]], Z["ba"][bad[:10]])


def test_float32_inputs_agree_almost_everywhere():
    """On the GPU path observations are float32: ties of argmin/thresholds may flip on a handful."""
    obs = torch.from_numpy(Z["obs"]).float()
    assert (apf_act_batch(obs, Z["a"], Z["w"]).numpy() != Z["apf"]).mean() < 0.01
    assert (ba_act_batch(obs, Z["a"], Z["w"]).numpy() != Z["ba"]).mean() < 0.01
