"""ReplayBuffer against the reference's own thirdparty/IQN/replay_buffer.py (golden G12, tests/golden/make_golden.py:
1500 `add`s into maxlen 1000, the deque contents afterwards, one `sample()`), on CPU tensors and -- through the HIP
append kernels (`mn_replay_append`, `mn_step_append`) -- on the GPU."""
import os

import numpy as np
import pytest
import torch

from distributional_rl_navigation_amd.iqn.replay_buffer import ReplayBuffer

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g12_replay.npz"))
CAP, B = int(Z["capacity"]), int(Z["batch"])


def _fifo(buf):
    """Ring contents oldest -> newest (the deque's iteration order)."""
    order = (torch.arange(buf.size, device=buf.device) + (buf.ptr if buf.size == buf.capacity else 0)) % buf.capacity
    return tuple(t[order].cpu().numpy() for t in (buf.states, buf.actions, buf.rewards, buf.next_states, buf.dones))


def _check_contents(buf):
    s, a, r, ns, d = _fifo(buf)
    assert len(buf) == len(Z["mem_states"]) == CAP
    assert np.array_equal(s, Z["mem_states"].astype(np.float32)) and np.array_equal(ns, Z["mem_next"].astype(np.float32))
    assert np.array_equal(a[:, 0], Z["mem_actions"]) and np.array_equal(r[:, 0], Z["mem_rewards"].astype(np.float32))
    assert np.array_equal(d[:, 0], Z["mem_dones"].astype(np.float32))


def _check_sample(buf):
    out = buf.sample()
    names = ("sample_states", "sample_actions", "sample_rewards", "sample_next", "sample_dones")
    for t, k in zip(out, names):      # shapes and dtypes of replay_buffer.py:49-57
        assert tuple(t.shape) == Z[k].shape and str(t.dtype).replace("torch.", "") == str(Z[k + "_dtype"]), k
    # uniform WITHOUT replacement from the memory (random.sample, replay_buffer.py:47): rows are distinct members
    mem = {tuple(row) for row in Z["mem_states"].astype(np.float32)}
    rows = [tuple(row) for row in out[0].cpu().numpy()]
    assert len(set(rows)) == B and all(r in mem for r in rows)
    ref_rows = [tuple(row) for row in Z["sample_states"]]
    assert len(set(ref_rows)) == B and all(r in mem for r in ref_rows)     # the reference's own sample obeys the same
    # consistent tuples: the sampled (s, a, r, s', d) belong together
    idx = {tuple(row): i for i, row in enumerate(Z["mem_states"].astype(np.float32))}
    k = [idx[r] for r in rows]
    assert np.array_equal(out[1].cpu().numpy()[:, 0], Z["mem_actions"][k])
    assert np.array_equal(out[3].cpu().numpy(), Z["mem_next"][k].astype(np.float32))


def test_single_adds_match_reference_deque():
    buf = ReplayBuffer(CAP, B, "cpu", seed=5, gamma=0.99)
    for i in range(len(Z["in_actions"])):
        buf.add(Z["in_states"][i], int(Z["in_actions"][i]), float(Z["in_rewards"][i]), Z["in_next"][i], bool(Z["in_dones"][i]))
        assert len(buf) == Z["sizes"][i]
    _check_contents(buf)
    _check_sample(buf)


def test_sampling_without_replacement_is_uniform():
    """`sample_indices` (no full permutation of the ring): distinct, in range, uniform inclusion frequencies."""
    buf = ReplayBuffer(5000, 64, "cpu", seed=1, gamma=0.99)
    buf.size = 5000
    hits = torch.zeros(5000)
    for _ in range(400):
        idx = buf.sample_indices(64)
        assert idx.unique().numel() == 64 and int(idx.min()) >= 0 and int(idx.max()) < 5000
        hits[idx] += 1
    # 25 600 draws over 5 000 rows: mean 5.12 per row, binomial spread
    assert abs(float(hits.mean()) - 5.12) < 1e-6 and float(hits.max()) < 20 and float((hits == 0).float().mean()) < 0.03
    buf.size = 100       # dense regime (size <= 4 b)
    assert buf.sample_indices(64).unique().numel() == 64


@pytest.mark.gpu
@pytest.mark.parametrize("chunks", [(300, 300, 300, 300, 300), (1500,), (7, 993, 500), (1000, 500)])
def test_vector_appends_match_reference_deque_on_device(chunks):
    """The same 1500 transitions appended in vector steps by the HIP kernel (`mn_replay_append`): the ring, read
    oldest -> newest, is the reference deque -- incl. wrap-around and a vector step larger than the capacity."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    dev = "cuda:0"
    buf = ReplayBuffer(CAP, B, dev, seed=5, gamma=0.99)
    lo = 0
    for n in chunks:
        sl = slice(lo, lo + n); lo += n
        buf.add_vector_step(torch.from_numpy(Z["in_states"][sl]).float().to(dev), torch.from_numpy(Z["in_actions"][sl]).to(dev, torch.int32),
                            torch.from_numpy(Z["in_rewards"][sl]).float().to(dev), torch.from_numpy(Z["in_next"][sl]).float().to(dev),
                            torch.from_numpy(Z["in_dones"][sl]).to(dev, torch.uint8))
        assert len(buf) == min(CAP, lo)
    _check_contents(buf)
    _check_sample(buf)


def test_n_step_returns_match_reference_deque():
    """n_step = 3 against the reference's own n-step ReplayBuffer (golden G15, replay_buffer.py:26-41): sizes after every add, the
    surviving transitions (first state / action of the window, discounted 3-step return, last next_state / done), the window sliding
    across episode ends as the reference's does; then the same stream as stream 0 of a 4-stream vector add."""
    Z15 = np.load(os.path.join(os.path.dirname(__file__), "golden", "g15_replay_nstep.npz"))
    cap, n_step, gamma = int(Z15["capacity"]), int(Z15["n_step"]), float(Z15["gamma"])
    buf = ReplayBuffer(cap, 8, "cpu", seed=5, gamma=gamma, n_step=n_step)
    for i in range(len(Z15["in_actions"])):
        buf.add(Z15["in_states"][i], int(Z15["in_actions"][i]), float(Z15["in_rewards"][i]), Z15["in_next"][i], bool(Z15["in_dones"][i]))
        assert len(buf) == Z15["sizes"][i]
    s, a, r, ns, d = _fifo(buf)
    assert np.array_equal(s, Z15["mem_states"].astype(np.float32)) and np.array_equal(ns, Z15["mem_next"].astype(np.float32))
    assert np.array_equal(a[:, 0], Z15["mem_actions"]) and np.array_equal(d[:, 0], Z15["mem_dones"].astype(np.float32))
    np.testing.assert_allclose(r[:, 0], Z15["mem_rewards"], rtol=0, atol=1e-6)
    # four parallel streams (a vector env): stream 0 carries the golden stream, the others shifted copies
    K = 4
    n_in = len(Z15["in_actions"])
    vb = ReplayBuffer(K * n_in, 8, "cpu", seed=5, gamma=gamma, n_step=n_step)
    for i in range(n_in):
        rows = [(i + 7 * k) % n_in for k in range(K)]
        vb.add_batch(torch.tensor(Z15["in_states"][rows], dtype=torch.float32), torch.tensor(Z15["in_actions"][rows]),
                     torch.tensor(Z15["in_rewards"][rows], dtype=torch.float32), torch.tensor(Z15["in_next"][rows], dtype=torch.float32),
                     torch.tensor(Z15["in_dones"][rows].astype(np.float32)))
    assert len(vb) == K * (n_in - n_step + 1)
    got = vb.rewards[0:len(vb):K, 0].numpy()[-cap:]          # stream 0's emissions, newest `cap`
    np.testing.assert_allclose(got, Z15["mem_rewards"], rtol=0, atol=1e-6)
    assert np.array_equal(vb.states[0:len(vb):K].numpy()[-cap:], Z15["mem_states"].astype(np.float32))
