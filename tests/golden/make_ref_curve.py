"""The reference's own learning curve as a small fixture: pretrained_models/IQN/seed_3/greedy_evaluations.npz (300 evaluations of the shipped model's training run,
30 evaluation worlds each) reduced to one row per evaluation -- timesteps, successes out of 30, mean cumulative reward.  Data of the reference's shipped artefact, no code.
Run in the build container (needs /root/reference): python tests/golden/make_ref_curve.py"""
import os
import numpy as np

SRC = "/root/reference/pretrained_models/IQN/seed_3/greedy_evaluations.npz"
d = np.load(SRC, allow_pickle=True)
t = np.asarray(d["timesteps"], dtype=np.int64)
succ = np.asarray(d["successes"]).astype(np.int64).sum(axis=1)
ret = np.asarray(d["rewards"], dtype=np.float64).mean(axis=1)
assert t.shape == succ.shape == ret.shape
out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_iqn_seed3_greedy_curve.npz")
np.savez(out, timesteps=t, successes=succ, mean_return=ret, n_worlds=np.int64(np.asarray(d["successes"]).shape[1]))
print(out, t.shape, "final", succ[-1], ret[-1], "best", succ.max(), ret.max())
