"""DQN baseline on the GPU: greedy policy over the HIP simulator against the checkpoint's recorded evaluation,
and the full 8-policy experiment sweep (run_experiments.py:213-218)."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")
TF32_GAP = 0.05     # see tests/test_dqn_cpu.py / make_golden_dqn.py


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


def test_dqn_q_values_on_device(torch):
    from distributional_rl_navigation_amd.dqn import DQNPolicy
    g = np.load(os.path.join(G, "g10_dqn.npz"))
    pol = DQNPolicy.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"), device="cuda:0")
    assert pol.use_fused_act      # the hand-written kernel (csrc/dqn_act.hip, exact-f32 MFMA) is the default on the GPU
    q = pol.q_values(torch.from_numpy(g["obs"]).cuda()).cpu().numpy()
    # G10's Q-values are the reference network's own float32 forward on the CPU, which is itself 5.3e-4 away from a float64 evaluation of
    # the same weights (|Q| up to 109: cancellation in the last layers).  So the bar is the float64 evaluation: the kernel's error must be
    # of the same size as the reference's own float32 forward's (observed: 5.8e-4 vs 5.3e-4), and it lies within the sum of both of the golden values
    import copy
    with torch.no_grad():
        q64 = copy.deepcopy(pol.q_net).double()(torch.from_numpy(g["obs"]).cuda().double()).cpu().numpy()
    err_golden, err_kernel = np.abs(g["q"] - q64).max(), np.abs(q - q64).max()
    assert err_kernel <= 1.5 * err_golden + 1e-5, (err_kernel, err_golden)
    np.testing.assert_allclose(q, g["q"], rtol=0, atol=err_golden + err_kernel + 1e-6)
    a = pol.act_batch(torch.from_numpy(g["obs"]).cuda()).cpu().numpy()
    top2 = np.sort(g["q"], axis=1)
    clear = (top2[:, -1] - top2[:, -2]) > 1e-3
    assert np.array_equal(a[clear], g["action"][clear]) and a.dtype == np.int32


def test_dqn_kernel_against_float64_and_eager_for_any_batch_size(torch):
    """`mn_dqn_act`: error against a float64 evaluation of the network no larger than eager PyTorch float32's, rows independent of the
    batch size / position (ragged tiles of 16), greedy action = first argmax of its own Q-values, weight changes picked up."""
    import copy
    from distributional_rl_navigation_amd.dqn import DQNPolicy
    pol = DQNPolicy.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"), device="cuda:0")
    gen = torch.Generator(device="cuda:0"); gen.manual_seed(5)
    obs = torch.randn(4099, 26, device="cuda:0", generator=gen) * 5.0
    obs[:, 4:][torch.rand(4099, 22, device="cuda:0", generator=gen) < 0.4] = 0.0
    with torch.no_grad():
        ref = copy.deepcopy(pol.q_net).double()(obs.double())
    q = pol.q_values(obs)
    pol.use_fused_act = False
    q_eager = pol.q_values(obs)
    pol.use_fused_act = True
    scale = float(ref.abs().max())
    e_hip, e_eager = float((q.double() - ref).abs().max()) / scale, float((q_eager.double() - ref).abs().max()) / scale
    # (this network is badly conditioned for float32 -- random observations give ~1e-5 of max |Q| in ANY float32 evaluation -- so the
    # yardstick is eager PyTorch float32 on the same GPU, not an absolute figure)
    assert e_hip < 1.5 * e_eager + 2e-7 and e_hip < 1e-4, (e_hip, e_eager)
    a = pol.act_batch(obs)
    assert a.dtype == torch.int32 and bool((a.long() == q.argmax(1)).all())
    for lo, hi in ((0, 1), (5, 22), (100, 116), (4000, 4099)):
        assert torch.equal(pol.q_values(obs[lo:hi]), q[lo:hi])
    with torch.no_grad():
        pol.q_net.q_net[2].weight.mul_(1.25); pol.q_net.features_extractor.hidden_layer.bias.add_(0.5)
        ref2 = copy.deepcopy(pol.q_net).double()(obs.double())
    q2 = pol.q_values(obs)
    pol.use_fused_act = False
    e2_eager = float((pol.q_values(obs).double() - ref2).abs().max()) / float(ref2.abs().max())
    pol.use_fused_act = True
    assert float((q2.double() - ref2).abs().max()) / float(ref2.abs().max()) < 1.5 * e2_eager + 2e-7 and not torch.equal(q2, q)


def test_dqn_closed_loop_on_hip_env(torch):
    """30 evaluation worlds side by side in the f64 HIP env, greedy DQN actions: per episode, the action sequence
    equals the recorded one up to the first step whose recorded action is within TF32_GAP of the max Q; fully
    reproduced episodes also reproduce the recorded discounted return and outcome."""
    from distributional_rl_navigation_amd.dqn import DQNPolicy
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    g = np.load(os.path.join(G, "g10_dqn.npz"))
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    pol = DQNPolicy.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"), device="cuda:0")
    env = VecMarineNavEnv(30, device="cuda:0", precision="f64", obs64=True)
    obs = env.load_worlds([VecMarineNavEnv.world_from_eval_config(c) for c in cfg.values()]).clone()
    rec, lens = g["eval_actions"].astype(np.int64), g["eval_len"]
    following = np.ones(30, dtype=bool)            # still on the recorded trajectory
    finished = np.zeros(30, dtype=bool)
    ret = np.zeros(30); outcome = np.zeros(30, dtype=np.int64)
    for t in range(int(lens.max())):
        q = pol.q_values(obs).cpu().numpy()
        a = q.argmax(1)
        for i in np.nonzero(following & ~finished)[0]:
            if a[i] != rec[i, t]:
                assert q[i].max() - q[i, rec[i, t]] < TF32_GAP, (i, t, q[i])
                following[i] = False
        act = np.where(following & ~finished, rec[:, min(t, rec.shape[1] - 1)], 0).clip(0, 8)
        obs, reward, done, info = env.step(torch.from_numpy(act.astype(np.int32)).cuda())
        r64 = env.get_reward64(); d = done.cpu().numpy().astype(bool); inf = info.cpu().numpy()
        live = following & ~finished
        ret[live] += 0.99 ** t * r64[live]
        for i in np.nonzero(live & d)[0]:
            assert t + 1 == lens[i], (i, t, lens[i])
            finished[i] = True; outcome[i] = inf[i]
        assert not np.any(live & ~d & (t + 1 >= lens)), "episode outlived its recording"
    full = following & finished
    assert full.sum() >= 15, full.sum()
    assert np.abs(ret[full] - g["eval_rewards"][full]).max() < 1e-4      # sb3 keeps rewards in float32
    assert np.array_equal(outcome[full] == 4, g["eval_successes"][full].astype(bool))
    env.close()


def test_full_policy_sweep_with_dqn(torch):
    from distributional_rl_navigation_amd.dqn import DQNPolicy
    from distributional_rl_navigation_amd.experiments import ALL_POLICIES, run_experiment
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    agent = IQNAgent(26, 9, device="cuda:0", seed=2, BUFFER_SIZE=1024)
    agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
    pol = DQNPolicy.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"), device="cuda:0")
    assert ALL_POLICIES == ("adaptive_IQN", "IQN_0.25", "IQN_0.5", "IQN_0.75", "IQN_1.0", "DQN", "APF", "BA")
    res, _ = run_experiment(agent, n_obs=6, n_cores=4, num=24, seed=15, policies=ALL_POLICIES, dqn=pol)
    assert list(res.keys()) == list(ALL_POLICIES)
    assert sum(res["DQN"]["success"]) >= 12, res["DQN"]["success"]
    with pytest.raises(ValueError):
        run_experiment(agent, n_obs=6, n_cores=4, num=2, policies=("DQN",))
    only, _ = run_experiment(None, n_obs=6, n_cores=4, num=24, seed=15, policies=("DQN",), dqn=pol)
    assert only["DQN"]["actions"] == res["DQN"]["actions"]          # rows are independent of the other policies


def test_dqn_learn_vec_and_checkpoint(torch, tmp_path):
    """DQN learner on the HIP vector env: cadence (updates every train_freq vector steps after learning_starts env steps,
    hard target copy every target_update_interval env steps), finite losses, weights move, and the checkpoint is an
    sb3-style policy.pth that `DQNPolicy.load` / `DQNAgent.load` read back."""
    from distributional_rl_navigation_amd.dqn import DQNAgent, DQNPolicy
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(2048, seed=0, device="cuda:0")
    ag = DQNAgent(device="cuda:0", buffer_size=50_000, batch_size=64, learning_starts=4096, train_freq=2,
                  target_update_interval=20480, seed=3)
    before = [p.detach().clone() for p in ag.q_net.parameters()]
    tgt0 = [p.detach().clone() for p in ag.q_net_target.parameters()]
    stats = ag.learn_vec(total_vector_steps=30, train_env=env)
    assert ag.num_timesteps == 30 * 2048 and len(ag.memory) == 50_000
    assert stats["n_updates"] == 14 and np.isfinite(stats["mean_loss"])      # vector steps 4, 6, ..., 30 (after 4096 env steps)
    assert all(float((p.detach() - q).abs().max()) > 0 for p, q in zip(ag.q_net.parameters(), before))
    # target copied at vector steps 10, 20, 30 (20480 env steps / 2048): equals the online net right after step 30's update order
    assert any(float((p.detach() - q).abs().max()) > 0 for p, q in zip(ag.q_net_target.parameters(), tgt0))
    ag.save(str(tmp_path))
    pol = DQNPolicy.load(os.path.join(tmp_path, "policy.pth"), device="cuda:0")
    obs = env.reset()
    assert torch.equal(pol.act_batch(obs), ag.policy.act_batch(obs))
    ag2 = DQNAgent(device="cuda:0", buffer_size=64, seed=9)
    ag2.load(os.path.join(tmp_path, "policy.pth"))
    for p, q in zip(ag.q_net_target.parameters(), ag2.q_net_target.parameters()):
        assert torch.equal(p, q)
    a = ag.act_batch(obs, 1.0)
    assert a.dtype == torch.int32 and int(a.min()) >= 0 and int(a.max()) <= 8 and a.unique().numel() == 9
    env.close()
