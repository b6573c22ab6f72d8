"""GPU integration tests: gym-shaped facade, create_eval_configs, batched evaluation with the reference's
pretrained checkpoint, and the vectorised learn loop."""
import json
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
G = os.path.join(os.path.dirname(__file__), "golden")


@pytest.fixture(scope="module")
def torch():
    import torch as t
    if not t.cuda.is_available():
        pytest.skip("no GPU")
    return t


def test_facade_follows_golden_trace(torch):
    """MarineNavEnv facade (n = 1, float64) replays the reference trace: reset/step return types and
    values, info strings, caller-side reset on done."""
    from distributional_rl_navigation_amd.marinenav_env.env import MarineNavEnv
    z = np.load(os.path.join(G, "g2_trace_seed0_default.npz"))
    env = MarineNavEnv(seed=int(z["seed"]))
    assert env.get_state_space_dimension() == 26 and env.get_action_space_dimension() == 9
    obs = env.reset()
    assert obs.dtype == np.float64 and obs.shape == (26,)
    np.testing.assert_allclose(obs, z["obs0"], atol=1e-10)
    names = ("normal", "out of boundary", "too long episode", "collision", "reach goal")
    for t in range(400):
        obs, r, d, info = env.step(int(z["actions"][t]))
        assert isinstance(r, float) and isinstance(d, bool) and info["state"] == names[int(z["info"][t])]
        np.testing.assert_allclose(obs, z["obs"][t], atol=1e-7)
        assert abs(r - z["reward"][t]) < 1e-7 and d == bool(z["done"][t])
        assert env.episode_timesteps == z["ep_t"][t] and env.total_timesteps == z["tot_t"][t]
        if d:
            ro = env.reset()
            np.testing.assert_allclose(ro, z["reset_obs"][t], atol=1e-10)
    ep = env.episode_data()
    assert set(ep.keys()) == {"env", "robot"} and len(ep["env"]["cores"]["positions"]) == len(env.cores)
    assert env.robot.compute_action_energy_cost(0) == 2.0 and env.robot.compute_action_energy_cost(4) == 0.0
    env.close()


def test_create_eval_configs_matches_reference_file(torch):
    from distributional_rl_navigation_amd.train_iqn import create_eval_configs
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        ref = json.load(f)
    cfg = create_eval_configs("cuda:0")
    assert list(cfg.keys()) == list(ref.keys())
    for k in ref:
        assert cfg[k]["env"]["cores"] == ref[k]["env"]["cores"], k
        assert cfg[k]["env"]["obstacles"] == ref[k]["env"]["obstacles"], k
        for f_ in ("init_theta", "init_speed", "dt", "N", "a", "w", "sonar"):
            assert cfg[k]["robot"][f_] == ref[k]["robot"][f_], (k, f_)
        for f_ in ("start", "goal", "width", "height", "r", "goal_dis", "discount", "seed"):
            assert cfg[k]["env"][f_] == ref[k]["env"][f_], (k, f_)


def test_pretrained_policy_batched_evaluation(torch, tmp_path):
    """The reference's trained IQN (pretrained_models/IQN/seed_3) driven through the batched env and
    act_batch: its stored final evaluation was 26/30 greedy, 25/30 adaptive; taus are random, so
    require >= 22/30 and the reference's npz schema."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    agent = IQNAgent(26, 9, device="cuda:0", seed=0, BUFFER_SIZE=1024)
    agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
    env = VecMarineNavEnv(30, device="cuda:0", precision="f64")
    for greedy in (True, False):
        res = agent.evaluation_vec(env, cfg, greedy=greedy, eval_log_path=str(tmp_path))
        assert sum(res["successes"]) >= 22, res["successes"]
        assert np.mean(res["rewards"]) > 40.0
    z = np.load(os.path.join(tmp_path, "greedy_evaluations.npz"), allow_pickle=True)
    assert sorted(z.files) == ["actions", "energies", "rewards", "successes", "times", "timesteps"]
    assert z["rewards"].shape == (1, 30) and z["actions"].shape == (1, 30)
    L = len(z["actions"][0][0])
    assert abs(z["times"][0][0] - 0.1 * 10 * L) < 1e-9
    env.close()


def test_learn_vec_runs_and_trains(torch):
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(1024, seed=0, device="cuda:0")
    agent = IQNAgent(26, 9, BATCH_SIZE=128, BUFFER_SIZE=20000, device="cuda:0", seed=1, learning_starts=2048,
                     target_update_interval=8)
    before = [p.detach().clone() for p in agent.qnetwork_local.parameters()]
    stats = agent.learn_vec(total_vector_steps=24, train_env=env, verbose=True)
    assert agent.current_timestep == 24 * 1024 and agent.learning_timestep == 22
    assert agent.grad_steps == 6 and len(agent.memory) == 20000
    assert stats["loss"] is not None and torch.isfinite(stats["loss"])
    assert any(not torch.equal(a, b) for a, b in zip(before, agent.qnetwork_local.parameters()))
    # replay holds consistent transitions: reward/done of the stored rows are finite / binary
    assert torch.isfinite(agent.memory.states).all() and set(agent.memory.dones.unique().tolist()) <= {0.0, 1.0}
    env.close()


def test_train_driver_end_to_end(torch, tmp_path):
    """train_iqn.run_trial (counterpart of train_IQN_model.py:74-121) on a tiny budget: writes the
    reference's per-trial files with the reference's schemas."""
    from distributional_rl_navigation_amd import train_iqn
    params = dict(agent="IQN", seed=2, total_timesteps=40_000, eval_freq=20_000, save_dir=str(tmp_path),
                  training_time="test")
    d = train_iqn.run_trial("cuda:0", params, n_envs=1024, batch=64, replay=20_000)
    files = sorted(os.listdir(d))
    for f in ("trial_config.json", "training_schedule.json", "eval_config.json", "greedy_evaluations.npz",
              "adaptive_evaluations.npz", "network_params.pth", "constructor_params.json"):
        assert f in files, (f, files)
    with open(os.path.join(d, "eval_config.json")) as f, open(os.path.join(G, "eval_config_seed3.json")) as g:
        mine, ref = json.load(f), json.load(g)
    assert mine["env_7"]["env"]["cores"] == ref["env_7"]["env"]["cores"]      # same 30 eval worlds as the reference
    z = np.load(os.path.join(d, "greedy_evaluations.npz"), allow_pickle=True)
    assert z["rewards"].shape[1] == 30 and len(z["timesteps"]) == 2
    # evaluation points are reported in reference-equivalent timesteps (fractions of the run), not clamped to every step
    assert 0 < z["timesteps"][0] < z["timesteps"][1] <= params["total_timesteps"]
    with open(os.path.join(d, "trial_config.json")) as f:
        plan = json.load(f)["batched"]
    # the learner budget is the reference's sample count: 40 000 / 4 gradient steps x 32 = 5 000 steps of batch 64
    assert plan["total_grad_steps"] == 5000 and plan["samples"] == plan["reference_samples"] == 320_000
    assert plan["vector_steps"] == 5000 and plan["eval_every_vector_steps"] == 2500
    from distributional_rl_navigation_amd.iqn.model import ObsEncoder
    net = ObsEncoder.load(d)
    assert sum(p.numel() for p in net.parameters()) == 35785


def test_cvar_experiment_sweep(torch):
    """Batched run_experiments.py (exp_setup_5 + IQN policies): the world sequence equals the reference
    RNG stream's (oracle, seed 15, fixed start/goal, no random pose), and the pretrained policy behaves."""
    from distributional_rl_navigation_amd.experiments import run_experiment, POLICIES
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from oracle.oracle import OracleEnv
    agent = IQNAgent(26, 9, device="cuda:0", seed=2, BUFFER_SIZE=1024)
    agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
    num = 40
    res, worlds = run_experiment(agent, n_obs=8, n_cores=6, num=num, seed=15)
    o = OracleEnv(15)
    o.set_flags(reset_start_and_goal=False, random_reset_state=False, set_boundary=True)
    o.set_start_goal([5.0, 5.0], [45.0, 45.0]); o.set_robot_N(5); o.set_world_size(6, 8, 25.0)
    for w in worlds:
        o.reset()
        ow = o.get_world()
        assert np.array_equal(w["cores"], ow["cores"]) and np.array_equal(w["obstacles"], ow["obstacles"])
        assert w["init_theta"] == np.pi / 4 and w["init_speed"] == 0.0
    assert list(res.keys()) == list(POLICIES)
    for name, r in res.items():
        assert len(r["success"]) == num and len(r["actions"]) == num
        assert all(abs(t - 0.5 * len(a)) < 1e-9 for t, a in zip(r["time"], r["actions"]))   # dt * N = 0.5 s
        assert not any(s and o_ for s, o_ in zip(r["success"], r["out_of_area"]))
    assert np.mean(res["IQN_1.0"]["success"]) > 0.6 and np.mean(res["adaptive_IQN"]["success"]) > 0.6
    assert np.mean(res["APF"]["success"]) + np.mean(res["BA"]["success"]) > 0.2      # classical baselines do reach goals


def test_reference_shaped_single_env_learn_loop(torch, tmp_path):
    """Drop-in check: the reference-shaped IQNAgent.learn (agent.py:94-173) drives the gym-shaped facade
    (reset / step / reset_with_eval_config / discount / robot.*) exactly as train_IQN_model.py would."""
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.env import make
    sched = dict(timesteps=[0, 1000000, 2000000], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10],
                 min_start_goal_dis=[30.0, 35.0, 40.0])
    train_env = make("marinenav_env:marinenav_env-v0", seed=0, schedule=sched)
    eval_env = make("marinenav_env:marinenav_env-v0", seed=348)
    with open(os.path.join(G, "eval_config_seed3.json")) as f:
        cfg = json.load(f)
    cfg = {k: cfg[k] for k in ("env_0", "env_11")}                      # two eval worlds keep the test short
    agent = IQNAgent(train_env.get_state_space_dimension(), train_env.get_action_space_dimension(), device="cuda:0",
                     seed=100, BATCH_SIZE=32, BUFFER_SIZE=5000, learning_starts=150, target_update_interval=50)
    agent.learn(total_timesteps=260, train_env=train_env, eval_env=eval_env, eval_config=cfg, eval_freq=100,
                eval_log_path=str(tmp_path), verbose=False)
    assert agent.current_timestep == 261 and agent.learning_timestep == 111       # `<=` loop, agent.py:113
    assert agent.grad_steps == 28                                                  # every 4th learning step
    z = np.load(os.path.join(tmp_path, "greedy_evaluations.npz"), allow_pickle=True)
    assert list(z["timesteps"]) == [150, 250] and z["rewards"].shape == (2, 2)    # first eval at learning_starts (A9)
    assert os.path.exists(os.path.join(tmp_path, "network_params.pth"))
    w = train_env._venv.get_worlds(0, 1)[0]
    assert w["n_cores"] <= 4 and w["n_obs"] <= 6                                   # curriculum stage 0
    train_env.close(); eval_env.close()


def test_facade_state_queries_and_attribute_writes(torch, capsys):
    """The remaining public surface of the reference class (marinenav_env.py:264-342, 422-465, 89-104; robot.py:28):
    get_velocity against the reference's own values (golden G5), check_collision / check_reach_goal / out_of_boundary /
    dist_to_goal consistent with what step() reports, get_observation (both forms), the curriculum print block, and
    `robot.dt` / `robot.N` writes reaching the device."""
    from distributional_rl_navigation_amd.marinenav_env.env import Core, MarineNavEnv
    z = np.load(os.path.join(G, "g5_velocity.npz"))
    env = MarineNavEnv(seed=0)
    for i in range(0, len(z["n"]), 7):
        n = int(z["n"][i])
        env.cores = [Core(c[0], c[1], int(c[2]), c[3]) for c in z["cores"][i][:n]]
        np.testing.assert_allclose(env.get_velocity(float(z["xy"][i][0]), float(z["xy"][i][1])), z["v"][i], rtol=0, atol=1e-12)
    # state queries agree with the step ladder along a trace
    z2 = np.load(os.path.join(G, "g2_trace_seed0_default.npz"))
    env = MarineNavEnv(seed=int(z2["seed"]))
    obs = env.reset()
    assert np.array_equal(env.get_observation(), obs)
    v_r, pts, g_r = env.get_observation(for_visualize=True)
    assert np.array_equal(v_r, obs[:2]) and np.array_equal(g_r, obs[2:4]) and pts.shape == (3, 11)
    assert np.array_equal(pts[:2].T.reshape(-1) * np.repeat(pts[2], 2), obs[4:])
    seen = set()
    for t in range(300):
        d_before = env.dist_to_goal()
        obs, r, done, info = env.step(int(z2["actions"][t]))
        state = info["state"]
        seen.add(state)
        assert env.check_collision() == (state == "collision")
        assert env.check_reach_goal() == (state == "reach goal")
        assert not env.out_of_boundary() or not (0 <= env.robot.x <= 50 and 0 <= env.robot.y <= 50)
        if state == "normal":
            assert abs(r - (-1.0 + d_before - env.dist_to_goal())) < 1e-9          # marinenav_env.py:220-222
        # robot.trajectory: one point per sub-step (marinenav_env.py:211-212), the last one = the pose after the step
        assert len(env.robot.trajectory) == 10 * len(env.robot.action_history)
        assert env.robot.trajectory[-1] == [env.robot.x, env.robot.y]
        if done:
            env.reset()
    assert "collision" in seen or "reach goal" in seen
    env.close()
    # robot.dt / robot.N writes (run_experiments.py:204 sets N = 5) change the integration on the device
    a, b = MarineNavEnv(seed=4), MarineNavEnv(seed=4)
    a.reset(); b.reset()
    b.robot.dt = 0.05; b.robot.N = 20                       # same simulated second, finer steps
    a.step(4); b.step(4)
    pa, pb = np.array([a.robot.x, a.robot.y]), np.array([b.robot.x, b.robot.y])
    assert 0 < np.linalg.norm(pa - pb) < 0.5
    assert b._venv.params.dt == 0.05 and b._venv.params.N == 20 and b.episode_data()["robot"]["dt"] == 0.05
    a.close(); b.close()
    # curriculum print block
    sched = dict(timesteps=[0, 5, 10], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10], min_start_goal_dis=[30.0, 35.0, 40.0])
    e = MarineNavEnv(seed=1, schedule=sched)
    capsys.readouterr()
    e.reset()
    out = capsys.readouterr().out
    assert "======== training schedule ========" in out and "num of cores:  4" in out and "min start goal dis:  30.0" in out
    for _ in range(6):
        e.step(0)
    e.reset()
    assert "num of cores:  6" in capsys.readouterr().out and e.num_cores == 6 and len(e.cores) <= 6
    e.close()


def test_act_eval_episodes_match_reference_closed_loop(torch):
    """Golden G14 = run_experiments.py's evaluation_IQN loop (:19-72) run with the reference env + the reference agent
    (pretrained seed_3, exp_setup_5 world 0 of seed 15, injected taus).  (1) Open loop: the stored observations through
    `act_eval_batch` give the stored quantiles / taus / CVaR levels / actions.  (2) Closed loop on the HIP env: the same
    injected taus reproduce the whole episode -- action sequence, sub-step trajectory, outcome, time, energy, return."""
    from distributional_rl_navigation_amd.experiments import _configure
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    Z = np.load(os.path.join(G, "g14_iqn_episodes.npz"))
    dev = "cuda:0"
    agent = IQNAgent(26, 9, device=dev, seed=2, BUFFER_SIZE=64)
    agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), dev)
    energy_tab = np.array([abs(a / 0.4) + abs(w / (np.pi / 6)) for a in (-0.4, 0.0, 0.4) for w in (-np.pi / 6, 0.0, np.pi / 6)])
    for name in ("adaptive", "cvar0.5", "cvar1.0"):
        obs_all = torch.from_numpy(Z[f"{name}_obs"]).float().to(dev)
        taus_in = torch.from_numpy(Z[f"{name}_taus_in"]).to(dev)
        T = len(obs_all)
        # (1) open loop
        cv = agent.adjust_cvar_batch(torch.from_numpy(Z[f"{name}_obs"]).to(dev)).float() if name == "adaptive" else \
            torch.full((T,), float(name[4:]), device=dev)
        np.testing.assert_allclose(cv.cpu().numpy(), Z[f"{name}_cvars"], rtol=1e-6, atol=1e-6)
        a, quant, taus = agent.act_eval_batch(obs_all, 0.0, cv, taus=taus_in)
        ref_q = Z[f"{name}_quantiles"][:, 0]
        np.testing.assert_allclose(quant.cpu().numpy(), ref_q, rtol=2e-5, atol=2e-4)
        np.testing.assert_allclose(taus.cpu().numpy(), Z[f"{name}_taus"][:, 0], rtol=0, atol=1e-7)
        top2 = np.sort(ref_q.mean(axis=1), axis=1)[:, -2:]
        clear = (top2[:, 1] - top2[:, 0]) > 1e-3
        assert clear.mean() > 0.9 and np.array_equal(a.cpu().numpy()[clear], Z[f"{name}_actions"][clear])
        # (2) closed loop
        env = VecMarineNavEnv(1, device=dev, precision="f64", obs64=True)
        _configure(env)
        env.enable_trajectory()
        obs = env.load_worlds([dict(cores=Z["world_cores"], obstacles=Z["world_obs"], start=[5.0, 5.0], goal=[45.0, 45.0],
                                    init_theta=np.pi / 4, init_speed=0.0)]).clone()
        np.testing.assert_allclose(env.get_obs64(0, 1)[0], Z["obs0"], atol=1e-9)
        acts, traj, ret, done, t = [], [], 0.0, False, 0
        while not done and t < 1000:
            cv_t = agent.adjust_cvar_batch(obs) if name == "adaptive" else float(name[4:])
            a_t, _, _ = agent.act_eval_batch(obs.contiguous(), 0.0, cv_t, taus=taus_in[t:t + 1]) if t < T else (None, None, None)
            assert a_t is not None, "episode outlived the reference's"
            obs, r, d, info = env.step(a_t)
            acts.append(int(a_t[0])); traj.extend(env.get_trajectory(0, 1)[0].tolist())
            ret += 0.99 ** t * float(env.get_reward64(0, 1)[0]); done = bool(d[0]); t += 1
        assert acts == list(Z[f"{name}_actions"]), name
        np.testing.assert_allclose(np.array(traj), Z[f"{name}_trajectory"], rtol=0, atol=1e-6)
        assert (int(info[0]) == 4) == bool(Z[f"{name}_success"]) and (int(info[0]) == 1) == bool(Z[f"{name}_out_of_area"])
        assert abs(0.1 * 5 * t - float(Z[f"{name}_time"])) < 1e-9 and abs(energy_tab[acts].sum() - float(Z[f"{name}_energy"])) < 1e-9
        assert abs(ret - float(Z[f"{name}_return"])) < 1e-5
        env.close()


def test_experiment_capture_schema(torch):
    """`run_experiment(capture=True)` emits the reference's exp_data entries (run_experiments.py:62-69,262-282): per
    episode an episode_data() dict with the reference's keys, the sub-step trajectory, and for IQN policies the
    per-action cvars / quantiles [1,32,9] / taus [1,32,1]; JSON-serialisable."""
    from distributional_rl_navigation_amd.experiments import run_experiment
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    Z = np.load(os.path.join(G, "g14_iqn_episodes.npz"))
    keys = json.loads(str(Z["adaptive_ep_keys"]))
    agent = IQNAgent(26, 9, device="cuda:0", seed=2, BUFFER_SIZE=64)
    agent.load_model(os.path.join(G, "pretrained_IQN_seed3"), "cuda:0")
    res, worlds = run_experiment(agent, n_obs=8, n_cores=6, num=6, seed=15, policies=("adaptive_IQN", "IQN_0.5", "APF"), capture=True)
    # world 0 of this sweep is the G14 world
    assert np.array_equal(worlds[0]["cores"], Z["world_cores"]) and np.array_equal(worlds[0]["obstacles"], Z["world_obs"])
    for name in ("adaptive_IQN", "IQN_0.5", "APF"):
        eps = res[name]["ep_data"]
        assert len(eps) == 6
        # exp_data[name]["computation_times"] (run_experiments.py:226,254-255): one entry per act call of the reference = per step of
        # every episode; here the amortised device time of the step's batched act launch
        ct = res[name]["computation_times"]
        assert len(ct) == sum(len(a_) for a_ in res[name]["actions"]) and all(0.0 < v < 1.0 for v in ct)
        assert set(res[name]) >= {"ep_data", "success", "time", "energy", "out_of_area", "computation_times"}      # run_experiments.py:226
        for i, ep in enumerate(eps):
            L = len(res[name]["actions"][i])
            assert sorted(ep["env"].keys()) == keys["env"]
            extra = {"actions_cvars", "actions_quantiles", "actions_taus"} if name != "APF" else set()
            assert set(ep["robot"].keys()) == set(keys["robot"]) | extra
            assert ep["robot"]["action_history"] == res[name]["actions"][i] and len(ep["robot"]["trajectory"]) == 5 * L
            assert ep["robot"]["N"] == 5 and ep["env"]["start"] == [5.0, 5.0] and ep["env"]["seed"] == 15
            if extra:
                assert len(ep["robot"]["actions_cvars"]) == L
                q = np.array(ep["robot"]["actions_quantiles"]); t = np.array(ep["robot"]["actions_taus"])
                assert q.shape == (L, 1, 32, 9) and t.shape == (L, 1, 32, 1)
                cv = np.array(ep["robot"]["actions_cvars"])
                assert (t[:, 0, :, 0].max(axis=1) <= cv + 1e-7).all()                  # taus are U[0,1) * cvar (model.py:149-153)
                if name == "IQN_0.5":
                    assert (cv == 0.5).all()
                assert np.array_equal(q[:, 0].mean(axis=1).argmax(axis=1), np.array(res[name]["actions"][i]))
    json.dumps(res)      # the reference dumps exp_data with json.dump
    # the G14 world with the reference's cvar = 0.5 policy ended out of area after 76 steps; with fresh taus the
    # batched run must at least reach the same kind of outcome record
    assert isinstance(res["IQN_0.5"]["out_of_area"][0], bool)


@pytest.mark.parametrize("precision,cvar,shared", [("f64", 1.0, None), ("mixed", 1.0, None), ("f64", 0.5, "collective"), ("f64", 0.5, "mailbox")],
                         ids=["f64", "mixed", "f64-cvar0.5-shared-collective", "f64-cvar0.5-shared-mailbox"])
def test_headline_configuration_loop_at_full_size(torch, precision, cvar, shared):
    """(`precision`: "f64" is what bench.py / train_iqn run when an IQN is in the loop -- `bench.default_precision` --, "mixed" the
    kernel-only configs' arithmetic.  The `shared` cases are ONE RANK's workload of BASELINE configs[4]: CVaR(0.5) action selection and
    `IQNAgent(distributed=True)` under an RCCL process group -- of one rank here, the box has one GPU -- with the gradient exchange
    as the RCCL all-reduce between the gradient and Adam launches, or over the mailbox inside the gradient step's one launch.)
    BASELINE configs[2] exactly as bench.py composes it -- 65 536 envs, replay 100 000, batch 256, one gradient step
    every 4 vector steps, fused act / step+append / reset / gradient-step kernels -- run for 24 vector steps with the
    bookkeeping and the data it leaves behind asserted at full size (size-independent properties)."""
    n, cap, B, T = 65536, 100_000, 256, 24
    dev = "cuda:0"
    if shared:
        import socket
        import torch.distributed as dist
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        with socket.socket() as sk:
            sk.bind(("127.0.0.1", 0))
            port = sk.getsockname()[1]
        dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1, device_id=torch.device(dev))
    try:
        _headline_loop(torch, precision, cvar, shared, n, cap, B, T, dev)
    finally:
        if shared:
            dist.destroy_process_group()


def _headline_loop(torch, precision, cvar, shared, n, cap, B, T, dev):
    from distributional_rl_navigation_amd.iqn.agent import IQNAgent
    from distributional_rl_navigation_amd.marinenav_env.vec_env import VecMarineNavEnv
    env = VecMarineNavEnv(n, seed=0, device=dev, precision=precision)
    env.set_attrs(num_cores=8, num_obs=10, min_start_goal_dis=40.0)
    agent = IQNAgent(26, 9, BATCH_SIZE=B, BUFFER_SIZE=cap, device=dev, seed=100, learning_starts=0, UPDATE_EVERY=4,
                     distributed=shared is not None)
    assert agent.use_fused_act and agent.use_fused_train
    if shared:
        agent.exchange = shared
    before = torch.cat([p.detach().reshape(-1).clone() for p in agent.qnetwork_local.parameters()])
    obs = env.reset()
    losses, dones = [], 0
    for t in range(T):
        prev = obs.clone()
        obs, reward, done, info, loss = agent.vec_step(env, obs, eps=0.5, cvar=cvar, per_iter=n)
        assert torch.isfinite(obs).all() and torch.isfinite(reward).all() and ((info != 0) == done.bool()).all()
        dones += int(done.sum())
        if loss is not None:
            losses.append(float(loss))
        # the transition block this step wrote: ring rows [ptr - n, ptr) hold (obs_t, a, r, ., done) of envs 0..n-1
        m = agent.memory
        lo = (m.ptr - n) % cap
        idx = (lo + torch.arange(n, device=dev)) % cap
        assert torch.equal(m.states[idx], prev) and torch.equal(m.rewards[idx, 0], reward) and torch.equal(m.dones[idx, 0], done.float())
        live = ~done.bool()
        assert torch.equal(m.next_states[idx][live], obs[live])           # finished envs: ring keeps the terminal observation,
        assert not torch.equal(m.next_states[idx][~live], obs[~live]) or int((~live).sum()) == 0   # `obs` already the new episode's first
        assert int(m.actions[idx].min()) >= 0 and int(m.actions[idx].max()) <= 8
    assert agent.current_timestep == T * n and agent.learning_timestep == T and agent.grad_steps == T // 4 == len(losses)
    assert len(agent.memory) == cap and agent.memory.ptr == (T * n) % cap
    assert all(np.isfinite(losses)) and dones > 0
    after = torch.cat([p.detach().reshape(-1) for p in agent.qnetwork_local.parameters()])
    assert bool(torch.isfinite(after).all()) and float((after - before).abs().max()) > 1e-5        # the learner moved the weights
    assert int(agent._fused.step_dev) == T // 4
    if shared:
        agent.check_learner()                                           # no bounded wait of the exchange ran out
        assert agent._fused.launches_per_step() == (1 if shared == "mailbox" else 3) and agent._fused.timeouts() == 0
    s, ep, tot = env.get_state()
    assert (tot == T).all() and (ep <= T).all()
    # exploration at eps = 0.5 with the library-drawn random numbers of that very call: greedy wherever u > eps, and the
    # greedy action is the argmax of the Q-values under the call's own taus
    from distributional_rl_navigation_amd.iqn.fused_act import fused_act
    a_mixed = agent.act_batch(obs, 0.5, cvar)
    d = agent._act_rng.draws(n, 32).clone()
    taus, u = d[:n * 32].view(n, 32), d[n * 32:]
    assert 0.9 * cvar < float(taus.max()) < cvar                                                # the draws ARE U[0,1) * cvar (model.py:149-153)
    a_greedy = fused_act(agent.qnetwork_local, obs.contiguous(), 0.0, 1.0, taus=taus)
    keep = u > 0.5
    assert torch.equal(a_mixed[keep], a_greedy[keep]) and 0.48 < float(keep.float().mean()) < 0.52
    hist = torch.bincount(a_mixed[~keep].long(), minlength=9).float()
    assert float((hist / hist.sum() - 1 / 9).abs().max()) < 0.01          # explored actions uniform over the 9
    env.close()
