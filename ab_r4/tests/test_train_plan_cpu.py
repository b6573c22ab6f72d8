"""`train_iqn.plan_cadence`: the reference's env-step cadences (train_IQN_model.py / agent.py defaults) translated to the
batched loop, and the driver's refusal of budgets that would not train."""
import pytest

from distributional_rl_navigation_amd.train_iqn import plan_cadence, trial_params


def test_headline_plan_keeps_the_reference_learner_budget():
    p = plan_cadence(3_000_000, 10_000, 65536, 256)
    assert p["reference_grad_steps"] == 750_000 and p["reference_samples"] == 24_000_000
    assert p["grad_steps_per_vector_step"] == 16
    assert p["total_grad_steps"] == p["vector_steps"] * 16 and abs(p["total_grad_steps"] - 93_750) < 16
    assert abs(p["samples"] - p["reference_samples"]) <= 16 * 256
    assert p["vector_steps"] == 5860 and p["n_evals"] == 30 and p["eval_every_vector_steps"] == 195      # never every step
    assert p["target_sync_grad_steps"] == 312                   # 2 500 reference grad steps x 32 / 256 samples
    assert abs(p["timestep_scale"] * p["vector_steps"] - 3_000_000) < 1e-6          # curriculum ends where the reference's does
    assert p["reference_replay_ratio"] == 8 and abs(p["replay_ratio"] - 16 * 256 / 65536) < 1e-12
    # 8 GPUs, shared learner: same learner budget, 8x the envs per vector step
    p8 = plan_cadence(3_000_000, 10_000, 8 * 65536, 256)
    assert p8["grad_steps_per_vector_step"] == 32 and abs(p8["total_grad_steps"] - 93_750) < 32
    # small batch of envs: one gradient step per vector step
    ps = plan_cadence(3_000_000, 10_000, 1024, 256)
    assert ps["grad_steps_per_vector_step"] == 1 and ps["vector_steps"] == 93_750


def test_explicit_overrides_and_grid():
    p = plan_cadence(3_000_000, 10_000, 65536, 256, grad_steps_per_vector_step=4, total_grad_steps=1000, n_evals=5)
    assert p["vector_steps"] == 250 and p["eval_every_vector_steps"] == 50 and p["total_grad_steps"] == 1000
    grid = trial_params(dict(agent="IQN", seed=[0, 1, 2], total_timesteps=3_000_000, eval_freq=10_000, save_dir="x"))
    assert [g["seed"] for g in grid] == [0, 1, 2] and all(g["agent"] == "IQN" for g in grid)      # train_IQN_model.py:52-65
    with pytest.raises(TypeError):
        trial_params(None)


def test_driver_refuses_a_budget_that_cannot_train(tmp_path):
    """Round 1's failure mode (46 vector steps, ~12 gradient steps for a 3 M-timestep config) is now an error, raised
    before anything touches the GPU."""
    from distributional_rl_navigation_amd.train_iqn import run_trial
    params = dict(agent="IQN", seed=0, total_timesteps=3_000_000, eval_freq=10_000, save_dir=str(tmp_path), training_time="t")
    with pytest.raises(ValueError, match="10x below"):
        run_trial("cuda:0", params, n_envs=65536, total_grad_steps=12)
