"""DQN learner on the vector env: the batched counterpart of the reference's DQN baseline training
(train_sb3_model.py:100-122 -> its modified stable-baselines3 `DQN`, thirdparty/stable_baselines3/dqn/dqn.py).

Same network (`ObsEncoderPolicy`, dqn/policies.py:212-240 = `dqn.policy.DQNPolicy` here), same update rule
(`DQN.train`, dqn.py:188-230: 1-step TD target from the target network, smooth-L1 loss, clip_grad_norm_(10), Adam 1e-4),
same schedules (`_on_step`, dqn.py:169-186: hard target copy every `target_update_interval` env steps, linear
exploration over the first `exploration_fraction` of training), driven by `VecMarineNavEnv` and the device replay ring
instead of DummyVecEnv (common/vec_env/dummy_vec_env.py:38-55: auto-reset, terminal observation kept for the
transition).  PyTorch-ROCm only -- this baseline is not on the north-star path and has no HIP kernels of its own.
"""
import copy
import os

import numpy as np
import torch
import torch.nn.functional as F

from ..iqn.replay_buffer import ReplayBuffer
from .policy import DQNPolicy


class DQNAgent:
    def __init__(self, state_size=26, action_size=9, learning_rate=1e-4, buffer_size=1_000_000, learning_starts=50000,
                 batch_size=32, tau=1.0, gamma=0.99, train_freq=4, gradient_steps=1, target_update_interval=10000,
                 exploration_fraction=0.1, exploration_initial_eps=1.0, exploration_final_eps=0.05, max_grad_norm=10,
                 device="cuda:0", seed=0):
        self.device = torch.device(device)
        torch.manual_seed(seed)                                    # sb3 set_random_seed (base_class.py) before the policy is built
        self.policy = DQNPolicy(state_size, action_size, device=device)
        self.q_net = self.policy.q_net
        self.q_net_target = copy.deepcopy(self.q_net)              # dqn/policies.py:150-156: target starts as a copy
        self.q_net_target.eval()
        self.optimizer = torch.optim.Adam(self.q_net.parameters(), lr=learning_rate)
        self.memory = ReplayBuffer(buffer_size, batch_size, device, seed, gamma, 1, state_size)
        self.batch_size, self.tau, self.gamma = batch_size, tau, gamma
        self.learning_starts, self.train_freq, self.gradient_steps = learning_starts, train_freq, gradient_steps
        self.target_update_interval, self.max_grad_norm = target_update_interval, max_grad_norm
        self.exploration_fraction = exploration_fraction
        self.exploration_initial_eps, self.exploration_final_eps = exploration_initial_eps, exploration_final_eps
        self.action_size = action_size
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) + 4321)
        self.num_timesteps = 0
        self.n_updates = 0

    # ---- update rule -----------------------------------------------------------------------------------------------
    def train(self, experiences=None):
        """One gradient step of DQN.train (dqn.py:196-224) on `experiences` = (obs, actions [B,1] i64, rewards [B,1],
        next_obs, dones [B,1] f32) or on a fresh sample of the replay ring.  Returns the loss (device scalar)."""
        obs, actions, rewards, next_obs, dones = experiences if experiences is not None else self.memory.sample()
        with torch.no_grad():
            next_q = self.q_net_target(next_obs).max(dim=1)[0].reshape(-1, 1)
            target_q = rewards + (1 - dones) * self.gamma * next_q
        current_q = torch.gather(self.q_net(obs), dim=1, index=actions.long())
        loss = F.smooth_l1_loss(current_q, target_q)
        self.optimizer.zero_grad()
        loss.backward()
        torch.nn.utils.clip_grad_norm_(self.q_net.parameters(), self.max_grad_norm)
        self.optimizer.step()
        self.n_updates += 1
        return loss.detach()

    def exploration_rate(self, total_timesteps):
        """get_linear_fn(initial, final, fraction) of the progress (common/utils.py), as dqn.py:120-124,183."""
        progress = self.num_timesteps / max(1, total_timesteps)
        if progress > self.exploration_fraction:
            return self.exploration_final_eps
        return self.exploration_initial_eps + progress * (self.exploration_final_eps - self.exploration_initial_eps) / self.exploration_fraction

    @torch.no_grad()
    def act_batch(self, obs, eps):
        """dqn.py:232-262 per env: a uniform random action with probability eps, else argmax_a Q."""
        greedy = self.policy.act_batch(obs)
        if eps <= 0.0:
            return greedy
        n = obs.shape[0]
        u = torch.rand(n, device=obs.device, generator=self.gen)
        rnd = torch.randint(0, self.action_size, (n,), device=obs.device, dtype=torch.int32, generator=self.gen)
        return torch.where(u < eps, rnd, greedy)

    # ---- vector loop -----------------------------------------------------------------------------------------------
    def learn_vec(self, total_vector_steps, train_env, total_timesteps=None, callback=None):
        """off_policy_algorithm.py:collect_rollouts / train cadence on n envs at once: every vector step adds n
        transitions; after `learning_starts` env steps, `gradient_steps` updates every `train_freq` VECTOR steps;
        target network copied every `target_update_interval` env steps (rounded to vector steps)."""
        n = train_env.n_envs
        total_timesteps = total_timesteps or total_vector_steps * n
        obs = train_env.reset()
        tgt_every = max(1, int(round(self.target_update_interval / n)))
        losses = []
        for it in range(total_vector_steps):
            eps = self.exploration_rate(total_timesteps)
            a = self.act_batch(obs, eps)
            nxt, reward, done, info = train_env.step(a)
            self.memory.add_vector_step(obs, a, reward, nxt, done)
            obs = train_env.reset_done()
            self.num_timesteps += n
            if (it + 1) % tgt_every == 0:                                       # dqn.py:175-176, tau = 1 -> hard copy
                with torch.no_grad():
                    for tp, lp in zip(self.q_net_target.parameters(), self.q_net.parameters()):
                        tp.mul_(1 - self.tau).add_(lp, alpha=self.tau)
            if self.num_timesteps > self.learning_starts and (it + 1) % self.train_freq == 0 and len(self.memory) >= self.batch_size:
                for _ in range(self.gradient_steps):
                    losses.append(self.train())
            if callback is not None:
                callback(self, it)
        return dict(vector_steps=total_vector_steps, n_updates=self.n_updates,
                    mean_loss=float(torch.stack(losses).mean()) if losses else float("nan"))

    # ---- checkpoints: the `policy.pth` of an sb3 zip (q_net.* and q_net_target.* keys) -----------------------------
    def state_dict(self):
        sd = {k: v.detach().clone() for k, v in self.policy.state_dict().items()}
        sd.update({"q_net_target." + k: v.detach().clone() for k, v in self.q_net_target.state_dict().items()})
        return sd

    def save(self, directory):
        os.makedirs(directory, exist_ok=True)
        torch.save(self.state_dict(), os.path.join(directory, "policy.pth"))

    def load(self, path):
        """`path`: an sb3 checkpoint zip, a policy.pth or the q_net npz fixture (target := q_net if absent)."""
        import io, zipfile
        if path.endswith(".npz"):
            sd = {k: torch.from_numpy(v) for k, v in np.load(path).items()}
        elif zipfile.is_zipfile(path) and "policy.pth" in zipfile.ZipFile(path).namelist():
            with zipfile.ZipFile(path) as z:
                sd = torch.load(io.BytesIO(z.read("policy.pth")), map_location="cpu")
        else:
            sd = torch.load(path, map_location="cpu")
        self.policy.load_state_dict({k: v for k, v in sd.items() if k.startswith("q_net.")}, strict=True)
        tgt = {k[len("q_net_target."):]: v for k, v in sd.items() if k.startswith("q_net_target.")}
        self.q_net_target.load_state_dict(tgt if tgt else self.q_net.state_dict())
