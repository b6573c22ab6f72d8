"""Training driver: counterpart of the reference's train_IQN_model.py (:15-179) for the batched path.

Same JSON config (`config/config_IQN.json` schema: agent, seed (list -> grid), total_timesteps,
eval_freq, save_dir), same per-trial outputs (`trial_config.json`, `training_schedule.json`,
`eval_config.json`, `greedy_evaluations.npz`, `adaptive_evaluations.npz`, `network_params.pth`,
`constructor_params.json`).  One process per GPU: launch with torch.distributed.run for several
GPUs (independent learners per rank by default, `--shared-learner` for one IQN with an RCCL
gradient all-reduce).

    python -m distributional_rl_navigation_amd.train_iqn -C config_IQN.json --n-envs 65536
"""
import argparse
import itertools
import json
import os
from datetime import datetime

import numpy as np


def trial_params(params):
    """train_IQN_model.py:52-65: list-valued keys expand to a Cartesian grid."""
    if isinstance(params, (str, int, float)):
        return [params]
    if isinstance(params, list):
        return params
    if isinstance(params, dict):
        keys, vals = zip(*params.items())
        return [dict(zip(keys, mix)) for mix in itertools.product(*[trial_params(v) for v in vals])]
    raise TypeError("Parameter type is incorrect.")


TRAINING_SCHEDULE = dict(timesteps=[0, 1000000, 2000000], num_cores=[4, 6, 8], num_obstacles=[6, 8, 10],
                         min_start_goal_dis=[30.0, 35.0, 40.0])   # train_IQN_model.py:86-90


def create_eval_configs(device, seed=348):
    """train_IQN_model.py:123-148: 30 evaluation worlds from ONE RNG stream (seed 348), fixed start/goal,
    10 x (4 cores, 6 obstacles), 10 x (6, 8), 10 x (8, 10).  Bit-identical to the reference's."""
    from .marinenav_env.env import MarineNavEnv
    env = MarineNavEnv(seed=seed, device=device)
    env.obs_r_range = [1, 3]
    env.reset_start_and_goal = False
    env.start = np.array([5.0, 5.0])
    env.goal = np.array([45.0, 45.0])
    cfg, count = {}, 0
    for nc, no in ((4, 6), (6, 8), (8, 10)):
        for _ in range(10):
            env.num_cores, env.num_obs = nc, no
            env.reset()
            cfg[f"env_{count}"] = env.episode_data()
            count += 1
    env.close()
    return cfg


def run_trial(device, params, n_envs, rank=0, world=1, shared=False, batch=256, replay=100_000, verbose=True,
              update_every=4, grad_steps=1, torch_train=False):
    """train_IQN_model.py:74-121 on the vector env."""
    import torch
    from .iqn.agent import IQNAgent
    from .marinenav_env.vec_env import VecMarineNavEnv

    exp_dir = os.path.join(params["save_dir"], "training_" + params["training_time"], "seed_" + str(params["seed"]))
    if world > 1 and not shared:
        exp_dir = os.path.join(exp_dir, f"rank_{rank}")
    writer = rank == 0 or not shared
    if writer:
        os.makedirs(exp_dir, exist_ok=True)
        with open(os.path.join(exp_dir, "trial_config.json"), "w+") as f:
            json.dump(params, f)
        with open(os.path.join(exp_dir, "training_schedule.json"), "w+") as f:
            json.dump(TRAINING_SCHEDULE, f)

    total = n_envs * world
    train_env = VecMarineNavEnv(n_envs, seed=params["seed"], first_index=rank * n_envs, schedule=TRAINING_SCHEDULE,
                                timestep_scale=total, device=device)
    eval_config = create_eval_configs(device)
    if writer:
        with open(os.path.join(exp_dir, "eval_config.json"), "w+") as f:
            json.dump(eval_config, f)
    eval_env = VecMarineNavEnv(len(eval_config), device=device) if writer else None

    agent = IQNAgent(26, 9, BATCH_SIZE=batch, BUFFER_SIZE=replay, device=device,
                     seed=params["seed"] + 100 + (0 if shared else rank), distributed=shared and world > 1,
                     UPDATE_EVERY=update_every)
    agent.grad_steps_per_update = grad_steps
    if torch_train:
        agent.use_fused_train = False          # PyTorch autograd + Adam instead of csrc/iqn_train.hip
    vec_steps = int(np.ceil((params["total_timesteps"] + 1) / total))
    eval_every = max(1, int(round(params["eval_freq"] / total)))
    agent.learn_vec(total_vector_steps=vec_steps, train_env=train_env, eval_env=eval_env, eval_config=eval_config,
                    eval_freq=eval_every, eval_log_path=exp_dir if writer else None,
                    total_timesteps=params["total_timesteps"], world_size=world, verbose=False)
    if writer:
        agent.qnetwork_local.save(exp_dir)
    train_env.close()
    if eval_env is not None:
        eval_env.close()
    torch.cuda.synchronize()
    return exp_dir


def main(argv=None):
    ap = argparse.ArgumentParser(description="Train IQN model (batched MI355X path)")
    ap.add_argument("-C", "--config-file", dest="config_file", type=open, required=True)
    ap.add_argument("-D", "--device", dest="device", type=str, default=None)
    ap.add_argument("--n-envs", type=int, default=65536, help="environments per GPU")
    ap.add_argument("--batch", type=int, default=256)
    ap.add_argument("--replay", type=int, default=100_000)
    ap.add_argument("--shared-learner", action="store_true")
    ap.add_argument("--update-every", type=int, default=4, help="vector steps between training events (UPDATE_EVERY)")
    ap.add_argument("--grad-steps", type=int, default=1, help="grad steps per training event")
    ap.add_argument("--torch-train", action="store_true", help="gradient step through PyTorch instead of the fused HIP kernels")
    args = ap.parse_args(argv)
    params = json.load(args.config_file)
    import torch
    world = int(os.environ.get("WORLD_SIZE", "1")); rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    device = args.device or f"cuda:{local}"
    torch.cuda.set_device(torch.device(device))
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device(device))
    stamp = datetime.now().strftime("%Y-%m-%d-%H-%M-%S")
    for p in trial_params(params):
        p["training_time"] = stamp
        run_trial(device, p, args.n_envs, rank, world, args.shared_learner, args.batch, args.replay,
                  update_every=args.update_every, grad_steps=args.grad_steps, torch_train=args.torch_train)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
