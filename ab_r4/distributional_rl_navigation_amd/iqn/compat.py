"""Single-env adapter: the reference's gym-shaped `IQNAgent` calls served by the batched machinery with n = 1.

`train_IQN_model.py` and `run_experiments.py` drive the agent with ONE gym-shaped env and numpy observation rows:
`learn`, `evaluation`, `act`, `act_eval`, `act_adaptive(_eval)`, `adjust_cvar`, `linear_eps` (thirdparty/IQN/agent.py:94-267,
319-398).  Nothing here owns network or cadence logic: a row becomes a [1, 26] tensor and goes through `qvals_batch` /
`act_eval_batch` / `adjust_cvar_batch` (`iqn/agent.py`: the fused HIP kernels on the GPU, PyTorch on the CPU), the loop asks
`cadence.cadence_tick` what is due, an episode is one call of `_play`.  What IS pinned here, because callers and the golden
G13 / G14 fixtures observe it: the exploration draw comes from python's `random` (one `random()` per action, one `choice` when it
explores -- agent.py:199-203), the `<=` loop bound, evaluation before the counter advances, the printed block and
the npz keys (`_log_evaluation`).
"""
import random

import numpy as np
import torch

from .cadence import cadence_tick

_TRAINING_INFO = ("current ep_length: ", "current ep_reward: ", "current ep_result: ", "episodes_num: ", "exploration_rate: ",
                  "current_timesteps: ", "total_timesteps: ")      # agent.py:158-168: labels of the per-episode print block


class _Episode:
    """Discounted return, length and energy of the episode being played."""

    def __init__(self, env):
        self.env, self.ret, self.steps, self.energy, self.actions, self.info = env, 0.0, 0, 0.0, [], {"state": "normal"}

    def advance(self, action):
        obs, reward, done, self.info = self.env.step(action)
        self.ret += reward * self.env.discount ** self.steps
        self.steps += 1
        return obs, reward, done

    def advance_logged(self, action):
        out = self.advance(action)
        self.energy += self.env.robot.compute_action_energy_cost(int(action))
        self.actions.append(int(action))
        return out


class ReferenceLoopMixin:
    # ---- row <-> batch ----------------------------------------------------------------------------------------------------
    def _row(self, state, dtype=torch.float32):
        return torch.as_tensor(np.asarray(state), dtype=dtype).reshape(1, -1).to(self.device)

    def _explore_or(self, greedy_action, eps):
        """agent.py:199-203 on python's `random`: greedy iff random() > eps, else a uniform choice over the action indices."""
        return int(greedy_action) if random.random() > eps else int(random.choice(np.arange(self.action_size)))

    def _quiet_net(self):
        class _Eval:
            def __enter__(s):
                self.qnetwork_local.eval()

            def __exit__(s, *exc):
                self.qnetwork_local.train()
        return _Eval()

    # ---- the reference's scalar helpers ------------------------------------------------------------------------------------
    def linear_eps(self, total_timesteps):
        """agent.py:176-183: linear ramp initial_eps -> final_eps over the first `exploration_fraction` of the run."""
        done_frac = self.current_timestep / total_timesteps
        if done_frac >= self.exploration_fraction:
            return self.final_eps
        return self.initial_eps + done_frac / self.exploration_fraction * (self.final_eps - self.initial_eps)

    def adjust_cvar(self, state):
        """agent.py:249-267 for one observation row (float64, like the reference's numpy arithmetic): min(1, closest return / 10)."""
        return float(self.adjust_cvar_batch(self._row(state, torch.float64))[0])

    # ---- acting on one row -------------------------------------------------------------------------------------------------
    def act(self, state, eps, cvar=1.0):
        """agent.py:186-205: one observation row -> python int."""
        with self._quiet_net():
            q = self.qvals_batch(self._row(state), cvar)
        return self._explore_or(q[0].argmax(), eps)

    def act_eval(self, state, eps=0.0, cvar=1.0):
        """agent.py:217-236: (action, quantiles [1, 32, 9], taus [1, 32, 1]) as numpy."""
        with self._quiet_net():
            greedy, quantiles, taus = self.act_eval_batch(self._row(state), 0.0, cvar)
        return self._explore_or(greedy[0], eps), quantiles.cpu().numpy(), taus.cpu().numpy()

    def act_adaptive(self, state, eps):
        """agent.py:207-215."""
        level = self.adjust_cvar(state)
        return self.act(state, eps, level), level

    def act_adaptive_eval(self, state, eps=0.0):
        """agent.py:238-247."""
        level = self.adjust_cvar(state)
        return self.act_eval(state, eps, level), level

    # ---- loops over a gym-shaped env ---------------------------------------------------------------------------------------
    def learn(self, total_timesteps, train_env, eval_env, eval_config, eval_freq, eval_log_path, verbose=True):
        """agent.py:94-173 with one gym-shaped env (the HIP facade `marinenav_env.env.MarineNavEnv` or any env of that shape)."""
        obs, ep, finished = train_env.reset(), _Episode(train_env), 0
        while self.current_timestep <= total_timesteps:
            eps = self.linear_eps(total_timesteps)
            action = self.act(obs, eps)
            nxt, reward, done = ep.advance(action)
            self.memory.add(obs, action, reward, nxt, done)
            obs = nxt
            due = cadence_tick(self, eval_freq=eval_freq)
            if due.train:
                self.train_from_memory()
            if due.sync:
                self._sync_target()
            if due.evaluate and eval_env is not None:
                for greedy in (True, False):
                    self.evaluation(eval_env, eval_config=eval_config, greedy=greedy, eval_log_path=eval_log_path)
                if eval_log_path is not None:
                    self.qnetwork_local.save(eval_log_path)
            if self.current_timestep >= self.learning_starts:
                self.learning_timestep += 1
            if done:
                finished += 1
                if verbose:
                    self._print_training_info((ep.steps, ep.ret, ep.info["state"], finished, eps, self.current_timestep, total_timesteps))
                obs, ep = train_env.reset(), _Episode(train_env)
            self.current_timestep += 1

    @staticmethod
    def _print_training_info(values):
        print("======== training info ========")
        for label, v in zip(_TRAINING_INFO, values):
            print(label, v)
        print("======== training info ========\n")

    def _play(self, env, first_obs, choose, max_steps=1000):
        """One evaluation episode: `choose(obs) -> action` until done or `max_steps` (agent.py:340-357)."""
        ep, obs, done = _Episode(env), first_obs, False
        while not done and ep.steps < max_steps:
            obs, _, done = ep.advance_logged(choose(obs))
        return ep

    def evaluation(self, eval_env, eval_config, greedy=True, eval_log_path=None):
        """agent.py:319-398 with one gym-shaped env: every evaluation world in turn, greedy or adaptive-CVaR policy."""
        choose = (lambda o: self.act(o, eps=0.0)) if greedy else (lambda o: self.act_adaptive(o, eps=0.0)[0])
        played = [self._play(eval_env, eval_env.reset_with_eval_config(world), choose) for world in eval_config.values()]
        seconds_per_step = eval_env.robot.dt * eval_env.robot.N
        self._log_evaluation(greedy, [e.actions for e in played], [e.ret for e in played], [e.info["state"] == "reach goal" for e in played],
                             [seconds_per_step * e.steps for e in played], [e.energy for e in played], eval_log_path)
