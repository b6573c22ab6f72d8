"""Reference-shaped single-env surface of the IQN agent -- the drop-in boundary layer, nothing more.

`train_IQN_model.py` and `run_experiments.py` drive `IQNAgent` through `learn`, `evaluation`, `act`, `act_eval`,
`act_adaptive(_eval)`, `adjust_cvar` and `linear_eps` with ONE gym-shaped env and numpy observations
(thirdparty/IQN/agent.py:94-267, 319-398).  Their bookkeeping (when `train` / `soft_update` / `evaluation` fire, what is
printed, which npz keys are written) is what those callers and the golden G13 / G14 fixtures pin, so these methods follow
the reference's control flow statement for statement.  They are kept apart from `iqn/agent.py` on purpose: everything
in THIS file is compatibility surface; the batched, device-resident code the MI355X path runs (`act_batch`, `vec_step`,
`learn_vec`, `evaluation_vec`, the fused HIP act / gradient kernels) is in `agent.py`, `fused_act.py`, `fused_train.py`.

The mixin expects the host class to provide: qnetwork_local, memory, device, action_size, current_timestep,
learning_timestep, learning_starts, UPDATE_EVERY, BATCH_SIZE, target_update_interval, exploration_fraction, initial_eps,
final_eps, train_from_memory(), soft_update(), _log_evaluation().
"""
import random

import numpy as np
import torch


class ReferenceLoopMixin:
    def linear_eps(self, total_timesteps):
        """agent.py:176-183."""
        progress = self.current_timestep / total_timesteps
        if progress < self.exploration_fraction:
            r = progress / self.exploration_fraction
            return self.initial_eps + r * (self.final_eps - self.initial_eps)
        return self.final_eps

    def adjust_cvar(self, state):
        """agent.py:249-267: cvar = min(1, closest sonar return / 10)."""
        sonar_points = state[4:]
        closest_d = np.inf
        for i in range(0, len(sonar_points), 2):
            x, y = sonar_points[i], sonar_points[i + 1]
            if np.abs(x) < 1e-3 and np.abs(y) < 1e-3:
                continue
            closest_d = min(closest_d, np.linalg.norm(sonar_points[i:i + 2]))
        cvar = 1.0
        if closest_d < 10.0:
            cvar = closest_d / 10.0
        return cvar

    def act(self, state, eps, cvar=1.0):
        """agent.py:186-205, one state (numpy) -> python int."""
        state = torch.from_numpy(np.asarray(state)).float().unsqueeze(0).to(self.device)
        self.qnetwork_local.eval()
        with torch.no_grad():
            action_values = self.qnetwork_local.get_qvals(state, cvar)
        self.qnetwork_local.train()
        if random.random() > eps:
            return int(np.argmax(action_values.cpu().data.numpy()))
        return int(random.choice(np.arange(self.action_size)))

    def act_adaptive(self, state, eps):
        cvar = self.adjust_cvar(state)
        return self.act(state, eps, cvar), cvar

    def act_eval(self, state, eps=0.0, cvar=1.0):
        """agent.py:217-236: action + the K quantiles and taus behind it."""
        state = torch.from_numpy(np.asarray(state)).float().unsqueeze(0).to(self.device)
        self.qnetwork_local.eval()
        with torch.no_grad():
            quantiles, taus = self.qnetwork_local.forward(state, self.qnetwork_local.K, cvar)
            action_values = quantiles.mean(dim=1)
        self.qnetwork_local.train()
        if random.random() > eps:
            action = int(np.argmax(action_values.cpu().data.numpy()))
        else:
            action = int(random.choice(np.arange(self.action_size)))
        return action, quantiles.cpu().data.numpy(), taus.cpu().data.numpy()

    def act_adaptive_eval(self, state, eps=0.0):
        cvar = self.adjust_cvar(state)
        return self.act_eval(state, eps, cvar), cvar

    def learn(self, total_timesteps, train_env, eval_env, eval_config, eval_freq, eval_log_path, verbose=True):
        """agent.py:94-173 with a gym-shaped single env (the facade MarineNavEnv or the reference's)."""
        state = train_env.reset()
        ep_reward, ep_length, ep_num = 0.0, 0, 0
        while self.current_timestep <= total_timesteps:
            eps = self.linear_eps(total_timesteps)
            action = self.act(state, eps)
            next_state, reward, done, info = train_env.step(action)
            ep_reward += train_env.discount ** ep_length * reward
            ep_length += 1
            self.memory.add(state, action, reward, next_state, done)
            state = next_state
            if self.current_timestep >= self.learning_starts:
                if self.learning_timestep % self.UPDATE_EVERY == 0 and len(self.memory) > self.BATCH_SIZE:
                    self.train_from_memory()
                if self.learning_timestep % self.target_update_interval == 0:
                    self.soft_update(self.qnetwork_local, self.qnetwork_target)
                if self.learning_timestep % eval_freq == 0 and eval_env is not None:
                    self.evaluation(eval_env, eval_config=eval_config, eval_log_path=eval_log_path)
                    self.evaluation(eval_env, eval_config=eval_config, greedy=False, eval_log_path=eval_log_path)
                    if eval_log_path is not None:
                        self.qnetwork_local.save(eval_log_path)
                self.learning_timestep += 1
            if done:
                ep_num += 1
                if verbose:
                    print("======== training info ========")
                    print("current ep_length: ", ep_length)
                    print("current ep_reward: ", ep_reward)
                    print("current ep_result: ", info["state"])
                    print("episodes_num: ", ep_num)
                    print("exploration_rate: ", eps)
                    print("current_timesteps: ", self.current_timestep)
                    print("total_timesteps: ", total_timesteps)
                    print("======== training info ========\n")
                ep_reward, ep_length = 0.0, 0
                state = train_env.reset()
            self.current_timestep += 1

    def evaluation(self, eval_env, eval_config, greedy=True, eval_log_path=None):
        """agent.py:319-398 with a gym-shaped single env."""
        action_data, reward_data, success_data, time_data, energy_data = [], [], [], [], []
        for idx, config in enumerate(eval_config.values()):
            observation = eval_env.reset_with_eval_config(config)
            actions, cumulative_reward, length, energy, done = [], 0.0, 0, 0.0, False
            info = {"state": "normal"}
            while not done and length < 1000:
                if greedy:
                    action = self.act(observation, eps=0.0)
                else:
                    action, _ = self.act_adaptive(observation, eps=0.0)
                observation, reward, done, info = eval_env.step(action)
                cumulative_reward += eval_env.discount ** length * reward
                length += 1
                energy += eval_env.robot.compute_action_energy_cost(int(action))
                actions.append(int(action))
            action_data.append(actions)
            reward_data.append(cumulative_reward)
            success_data.append(info["state"] == "reach goal")
            time_data.append(eval_env.robot.dt * eval_env.robot.N * length)
            energy_data.append(energy)
        self._log_evaluation(greedy, action_data, reward_data, success_data, time_data, energy_data, eval_log_path)
