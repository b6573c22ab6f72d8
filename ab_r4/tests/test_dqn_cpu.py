"""DQN baseline policy against golden vectors from the reference's own sb3 network
(tests/golden/make_golden_dqn.py) - CPU."""
import json
import os

import numpy as np
import torch

G = os.path.join(os.path.dirname(__file__), "golden")
TF32_GAP = 0.05     # the authors' recorded greedy actions came from a TF32-era GPU forward: see make_golden_dqn.py


def _policy():
    from distributional_rl_navigation_amd.dqn import DQNPolicy
    return DQNPolicy.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"), device="cpu")


def test_q_values_match_reference_network():
    g = np.load(os.path.join(G, "g10_dqn.npz"))
    pol = _policy()
    q = pol.q_values(torch.from_numpy(g["obs"])).numpy()
    np.testing.assert_allclose(q, g["q"], rtol=0, atol=1e-5)          # same f32 ops; tolerance = f32 rounding of |Q| ~ 50
    assert np.array_equal(pol.act_batch(torch.from_numpy(g["obs"])).numpy(), g["action"].astype(np.int32))
    a, state = pol.predict(g["obs"][0].astype(np.float64), deterministic=True)       # sb3 surface, run_experiments.py:86
    assert state is None and int(a) == int(g["action"][0])
    a, _ = pol.predict(g["obs"][:7])
    assert a.shape == (7,) and np.array_equal(a, g["action"][:7])


def test_stochastic_predict_is_sb3s_epsilon_greedy():
    """DQN.predict(deterministic=False) (dqn/dqn.py:249-257): one uniform draw per CALL; below exploration_rate every row gets a
    uniformly random action, otherwise the greedy actions."""
    g = np.load(os.path.join(G, "g10_dqn.npz"))
    pol = _policy()
    np.random.seed(0)
    greedy = g["action"][:64]
    explored, hist = 0, np.zeros(9)
    for _ in range(400):
        a, state = pol.predict(g["obs"][:64], deterministic=False)
        assert state is None and a.shape == (64,) and a.min() >= 0 and a.max() < 9
        if not np.array_equal(a, greedy):
            explored += 1
            hist += np.bincount(a, minlength=9)
    assert 8 <= explored <= 36                     # Binomial(400, 0.05): mean 20
    assert (hist / hist.sum()).max() < 0.16        # random rows are uniform over the 9 actions
    pol.exploration_rate = 0.0
    assert np.array_equal(pol.predict(g["obs"][:64], deterministic=False)[0], greedy)
    a, _ = pol.predict(g["obs"][0], deterministic=False)
    assert np.ndim(a) == 0


def test_state_dict_names_are_sb3s():
    keys = set(_policy().state_dict().keys())
    z = np.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"))
    assert keys == set(z.files) and len(keys) == 18
    assert "q_net.features_extractor.sensor_encoder.weight" in keys and "q_net.q_net.4.bias" in keys


def test_closed_loop_reproduces_recorded_evaluation():
    """Greedy episodes on the 30 evaluation worlds (oracle env as the CPU stand-in for the simulator) against the
    checkpoint's own evaluations.npz: an episode either replays the recorded action sequence AND discounted
    return exactly, or first departs from it at a step where the recorded action's Q is within TF32_GAP of the
    maximum."""
    from oracle.oracle import OracleEnv
    g = np.load(os.path.join(G, "g10_dqn.npz"))
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    pol = _policy()
    full = 0
    for i in range(30):
        env = OracleEnv(seed=0)
        obs = env.load_eval_config(cfg[f"env_{i}"])
        rec = g["eval_actions"][i][: g["eval_len"][i]]
        ret, done, t = 0.0, False, 0
        while not done and t < len(rec):
            q = pol.q_values(torch.as_tensor(np.asarray(obs), dtype=torch.float32))[0].numpy()
            if int(q.argmax()) != int(rec[t]):
                assert q.max() - q[int(rec[t])] < TF32_GAP, (i, t, q)
                break
            obs, r, done, info = env.step(int(rec[t]))
            ret += 0.99 ** t * r
            t += 1
        else:
            assert done and t == len(rec), (i, t)
            # sb3's DummyVecEnv keeps per-step rewards in a float32 buffer: 1e-4 on a discounted return of ~90
            assert abs(ret - g["eval_rewards"][i]) < 1e-4, (i, ret, g["eval_rewards"][i])
            assert (info == 4) == bool(g["eval_successes"][i])          # 4 = "reach goal"
            full += 1
    assert full >= 15, full


def test_train_step_matches_reference_dqn_train():
    """G11: one step of the reference's own `DQN.train` (dqn/dqn.py:188-230, run on its own ObsEncoderPolicy by
    tests/golden/make_golden_dqn.py) -> loss, clipped gradients and post-Adam parameters of `DQNAgent.train`."""
    from distributional_rl_navigation_amd.dqn import DQNAgent
    Z = np.load(os.path.join(G, "g11_dqn_train.npz"))
    ag = DQNAgent(device="cpu", buffer_size=64, batch_size=32)
    ag.load(os.path.join(G, "pretrained_DQN_seed3", "q_net.npz"))
    ag.q_net_target.load_state_dict({k[len("tgt_"):]: torch.from_numpy(Z[k]) for k in Z.files if k.startswith("tgt_")})
    exp = tuple(torch.from_numpy(Z["batch_" + k]) for k in ("observations", "actions", "rewards", "next_observations", "dones"))
    loss = ag.train(exp)
    np.testing.assert_allclose(float(loss), float(Z["loss"]), rtol=1e-6)
    for k, p in ag.q_net.named_parameters():
        np.testing.assert_allclose(p.grad.numpy(), Z["grad_" + k], rtol=1e-4, atol=1e-6, err_msg=k)
        np.testing.assert_allclose(p.detach().numpy(), Z["after_" + k], rtol=0, atol=2e-6, err_msg=k)
    assert ag.n_updates == 1
    sd = ag.state_dict()
    assert len(sd) == 36 and "q_net_target.q_net.4.bias" in sd and "q_net.features_extractor.goal_encoder.weight" in sd


def test_exploration_schedule_is_sb3s_linear_fn():
    from distributional_rl_navigation_amd.dqn import DQNAgent
    ag = DQNAgent(device="cpu", buffer_size=64)
    for t, want in ((0, 1.0), (150_000, 0.525), (300_000, 0.05), (2_000_000, 0.05)):
        ag.num_timesteps = t
        assert abs(ag.exploration_rate(3_000_000) - want) < 1e-12
