"""The drop-in `IQNAgent.learn` loop against the bookkeeping of the reference's own loop (thirdparty/IQN/agent.py:94-173)
recorded by running the reference agent on the reference env (golden G13, tests/golden/make_golden.py): counters,
the learning steps at which train() / soft_update() / evaluation() fire, the evaluation npz.  These are
trajectory-independent, so they must match exactly whatever the policy does."""
import json
import os

import numpy as np
import pytest
import torch

from distributional_rl_navigation_amd.iqn.agent import IQNAgent

Z = np.load(os.path.join(os.path.dirname(__file__), "golden", "g13_learn_loop.npz"), allow_pickle=True)
CFG = json.loads(str(Z["cfg"]))


class _FakeRobot:
    dt, N = 0.1, 10

    def compute_action_energy_cost(self, a):
        return 1.0


class _FakeEnv:
    """Duck-typed gym env: random observations, episodes of 40 steps."""
    discount = 0.99

    def __init__(self, seed):
        self.rng = np.random.RandomState(seed)
        self.robot = _FakeRobot()
        self.t = 0

    def reset(self):
        self.t = 0
        return self.rng.normal(0, 3, 26)

    def reset_with_eval_config(self, cfg):
        return self.reset()

    def step(self, a):
        self.t += 1
        done = self.t >= 40
        return self.rng.normal(0, 3, 26), float(self.rng.normal()), done, {"state": "too long episode" if done else "normal"}


def _run(agent, train_env, eval_env, eval_config, tmp_path):
    log = dict(train_at=[], train_mem=[], sync_at=[], eval_at=[], eval_ts=[])
    real_train, real_sync, real_eval = agent.train_from_memory, agent.soft_update, agent.evaluation

    def train():
        log["train_at"].append(agent.learning_timestep); log["train_mem"].append(len(agent.memory))
        return real_train()

    def sync(a, b):
        log["sync_at"].append(agent.learning_timestep)
        return real_sync(a, b)

    def evaluation(env, eval_config, greedy=True, eval_log_path=None):
        log["eval_at"].append((agent.learning_timestep, int(greedy))); log["eval_ts"].append(agent.current_timestep)
        return real_eval(env, eval_config=eval_config, greedy=greedy, eval_log_path=eval_log_path)

    agent.train_from_memory, agent.soft_update, agent.evaluation = train, sync, evaluation
    agent.learn(total_timesteps=CFG["total_timesteps"], train_env=train_env, eval_env=eval_env, eval_config=eval_config,
                eval_freq=CFG["eval_freq"], eval_log_path=str(tmp_path), verbose=False)
    return log


def _check(agent, log, tmp_path):
    assert agent.current_timestep == int(Z["current_timestep"]) and agent.learning_timestep == int(Z["learning_timestep"])
    assert log["train_at"] == list(Z["train_at"]) and log["train_mem"] == list(Z["train_mem"])
    assert agent.grad_steps == len(Z["train_at"])
    assert log["sync_at"] == list(Z["sync_at"])
    assert [list(x) for x in log["eval_at"]] == [list(x) for x in Z["eval_at"]] and log["eval_ts"] == list(Z["eval_ts"])
    assert len(agent.memory) == int(Z["memory_len"])
    assert sorted(os.listdir(tmp_path)) == list(Z["files"])
    zg = np.load(os.path.join(tmp_path, "greedy_evaluations.npz"), allow_pickle=True)
    za = np.load(os.path.join(tmp_path, "adaptive_evaluations.npz"), allow_pickle=True)
    assert sorted(zg.files) == list(Z["npz_keys"])
    assert list(zg["timesteps"]) == list(Z["greedy_timesteps"]) and list(za["timesteps"]) == list(Z["adaptive_timesteps"])
    assert list(zg["rewards"].shape) == list(Z["greedy_rewards_shape"]) and list(zg["successes"].shape) == list(Z["greedy_successes_shape"])


def _agent(device):
    return IQNAgent(26, 9, BATCH_SIZE=CFG["BATCH_SIZE"], BUFFER_SIZE=CFG["BUFFER_SIZE"], UPDATE_EVERY=CFG["UPDATE_EVERY"],
                    learning_starts=CFG["learning_starts"], target_update_interval=CFG["target_update_interval"],
                    seed=CFG["seed"], device=device)


def test_learn_loop_bookkeeping_cpu(tmp_path):
    agent = _agent("cpu")
    log = _run(agent, _FakeEnv(0), _FakeEnv(1), {"env_0": {}, "env_1": {}}, tmp_path)
    _check(agent, log, tmp_path)


@pytest.mark.gpu
def test_learn_loop_bookkeeping_on_device_facade(tmp_path):
    """The same loop driving the gym-shaped HIP facade (train env seeded like the reference run, the reference's two
    evaluation worlds), fused HIP act + gradient step: identical bookkeeping, identical env counters."""
    if not torch.cuda.is_available():
        pytest.skip("no GPU")
    from distributional_rl_navigation_amd.marinenav_env.env import MarineNavEnv
    agent = _agent("cuda:0")
    train_env = MarineNavEnv(seed=CFG["env_seed"])
    eval_env = MarineNavEnv(seed=348)
    log = _run(agent, train_env, eval_env, json.loads(str(Z["eval_config"])), tmp_path)
    _check(agent, log, tmp_path)
    assert train_env.total_timesteps == int(Z["env_total_timesteps"])
    train_env.close(); eval_env.close()
