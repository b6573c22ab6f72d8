"""iqn/overlap.py: the vectorised loop over two half batches on two HIP streams must be the same computation as the one-batch loop."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("needs a GPU")
    return t


def _mk(n, first, precision="f64"):
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    e = VecMarineNavEnv(n, seed=3, first_index=first, device="cuda:0", precision=precision)
    e.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    return e


def test_two_half_batches_on_two_streams_equal_the_one_batch_loop(torch):
    """20 vector steps of SplitBatchLoop (2 x 1024 envs, free-running streams, training events every 2 steps) -- then the SAME actions
    (read back from the replay ring, sub-batch 0's rows then sub-batch 1's per step) through ONE 2048-env handle with the plain
    step_append / reset_done pair: both replay rings (states, actions, rewards, next states, dones) and the final observations
    are bit-identical, i.e. sharding, the concurrent appends and the resets lose or tear nothing."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.iqn.overlap import SplitBatchLoop
    from distributional_rl_navigation_amd.iqn.replay_buffer import ReplayBuffer
    dev = "cuda:0"
    half, T = 1024, 20
    n = 2 * half
    ag = IQNAgent(26, 9, BATCH_SIZE=64, BUFFER_SIZE=T * n, device=dev, seed=5, learning_starts=0, UPDATE_EVERY=2)
    loop = SplitBatchLoop(ag, [_mk(half, 0), _mk(half, half)], act_grid=64)
    loop.reset()
    losses = []
    for t in range(T):
        out = loop.step(0.3, 1.0)
        if out[4] is not None:
            losses.append(out[4])
    loop.close(); torch.cuda.synchronize()
    assert ag.grad_steps == T // 2 and len(ag.memory) == T * n and all(np.isfinite(float(l)) for l in losses)
    final = torch.cat([o.clone() for o in loop.obs])
    # replay through one handle
    ref_env = _mk(n, 0)
    ref_mem = ReplayBuffer(T * n, 64, dev, 0, 0.99)
    obs = ref_env.reset()
    for t in range(T):
        a = ag.memory.actions[t * n:(t + 1) * n, 0].to(torch.int32).contiguous()
        ref_env.step_append(a, obs, ref_mem)
        obs = ref_env.reset_done()
    torch.cuda.synchronize()
    for name in ("states", "actions", "rewards", "next_states", "dones"):
        assert torch.equal(getattr(ag.memory, name), getattr(ref_mem, name)), name
    assert torch.equal(final, obs)
    assert int(ag.memory.dones.sum()) > 0          # episodes ended and were reset inside the window
    # the act path after the last training event sees the trained weights (image refreshed on the calling stream)
    from distributional_rl_navigation_amd.iqn.fused_act import fused_qvals
    taus = torch.rand(64, 32, device=dev)
    q_hip = fused_qvals(ag.qnetwork_local, final[:64].contiguous(), 1.0, taus=taus)
    q_ref = ag.qnetwork_local.get_qvals(final[:64], 1.0, taus=taus)
    assert float((q_hip - q_ref).abs().max()) <= 1e-4 * max(1.0, float(q_ref.abs().max()))
    for e in loop.envs:
        e.close()
    ref_env.close()


def test_one_sub_batch_is_the_plain_loop(torch):
    """H = 1 degenerates to vec_step's launch sequence on a side stream: counters, ring size and training cadence match."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.iqn.overlap import SplitBatchLoop
    ag = IQNAgent(26, 9, BATCH_SIZE=32, BUFFER_SIZE=4096, device="cuda:0", seed=1, learning_starts=0, UPDATE_EVERY=4)
    loop = SplitBatchLoop(ag, [_mk(256, 0, "mixed")])
    loop.reset()
    for _ in range(9):
        loop.step(1.0)
    loop.join(); torch.cuda.synchronize()
    assert ag.learning_timestep == 9 and ag.current_timestep == 9 * 256 and ag.grad_steps == 3 and len(ag.memory) == 9 * 256
    loop.envs[0].close()
