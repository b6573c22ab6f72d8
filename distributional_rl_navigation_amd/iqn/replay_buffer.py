"""Device-resident replay memory (counterpart of thirdparty/IQN/replay_buffer.py:6-59).

The reference keeps a deque of python tuples and stacks a batch on every sample; here the
transitions live in HBM as a ring of tensors and a whole vector step (n_envs transitions) is
appended with one indexed copy.

n_step > 1 (replay_buffer.py:26-41; never used by the reference's scripts, part of the constructor surface): every
environment stream keeps its last n_step transitions; once it holds n_step of them each further add emits
(s_{t-n+1}, a_{t-n+1}, sum_i gamma^i r_{t-n+1+i}, s'_t, done_t) -- the reference's sliding window, which (as there) does
not restart at episode ends.  Torch ops on the ring's device; the fused `mn_step_append` path stores 1-step transitions
only, so the agent uses step + `add_vector_step` when n_step > 1.
"""
import torch


class ReplayBuffer:
    def __init__(self, buffer_size, batch_size, device, seed, gamma, n_step=1, state_size=26):
        if int(n_step) < 1:
            raise ValueError("n_step must be >= 1")
        self.device = torch.device(device)
        self.capacity = int(buffer_size)
        self.batch_size = int(batch_size)
        self.gamma = gamma
        self.n_step = n_step
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed))
        c = self.capacity
        self.states = torch.zeros(c, state_size, dtype=torch.float32, device=self.device)
        self.next_states = torch.zeros(c, state_size, dtype=torch.float32, device=self.device)
        self.actions = torch.zeros(c, 1, dtype=torch.int64, device=self.device)
        self.rewards = torch.zeros(c, 1, dtype=torch.float32, device=self.device)
        self.dones = torch.zeros(c, 1, dtype=torch.float32, device=self.device)
        self.size = 0
        self.ptr = 0
        self._hist = None     # n_step > 1: per-stream window of the last n_step transitions
        self.version = 0      # bumped by every write to the ring (a batch the gradient kernels staged from it is then stale)

    def add(self, state, action, reward, next_state, done):
        """One transition (replay_buffer.py:26-34)."""
        as_t = lambda x, dt: torch.as_tensor(x, dtype=dt, device=self.device)
        self.add_batch(as_t(state, torch.float32).view(1, -1), as_t([action], torch.int64),
                       as_t([reward], torch.float32), as_t(next_state, torch.float32).view(1, -1),
                       as_t([float(done)], torch.float32))

    def _nstep_window(self, states, actions, rewards, next_states, dones):
        """replay_buffer.py:26-41 for `k` parallel streams (row i of every call belongs to stream i): returns the k n-step
        transitions this add completes, or None while the windows are still filling."""
        k = states.shape[0]
        n = self.n_step
        h = self._hist
        if h is None or h["s"].shape[1] != k:
            h = self._hist = dict(s=torch.zeros(n, k, states.shape[1], dtype=torch.float32, device=self.device),
                                  a=torch.zeros(n, k, dtype=torch.int64, device=self.device),
                                  r=torch.zeros(n, k, dtype=torch.float32, device=self.device), count=0)
        slot = h["count"] % n
        h["s"][slot].copy_(states); h["a"][slot].copy_(actions.view(-1)); h["r"][slot].copy_(rewards.view(-1))
        h["count"] += 1
        if h["count"] < n:
            return None
        oldest = h["count"] % n                     # the slot the NEXT add overwrites = the window's first transition
        ret = torch.zeros(k, dtype=torch.float32, device=self.device)
        for i in range(n):                          # Return += gamma**idx * reward_idx, in the reference's order
            ret = ret + (self.gamma ** i) * h["r"][(oldest + i) % n]
        return h["s"][oldest].clone(), h["a"][oldest].clone(), ret, next_states, dones

    def add_vector_step(self, obs, actions_i32, reward, next_obs, done_u8):
        """One vector step of the HIP env (device tensors exactly as VecMarineNavEnv returns them:
        obs / next_obs [n,26] f32, actions [n] i32, reward [n] f32, done [n] u8) appended by ONE kernel
        (csrc/replay.hip); same FIFO semantics as add_batch."""
        if self.n_step > 1:
            return self.add_batch(obs, actions_i32.long(), reward, next_obs, done_u8.float())
        import ctypes as C
        from .. import _capi
        n = obs.shape[0]
        p = lambda t: C.c_void_p(t.data_ptr())
        for t in (obs, actions_i32, reward, next_obs, done_u8):
            assert t.is_cuda and t.is_contiguous()
        assert actions_i32.dtype == torch.int32 and done_u8.dtype == torch.uint8 and obs.dtype == torch.float32
        stream = _capi.stream_ptr(self.device)
        rc = _capi.lib().mn_replay_append(p(obs), p(actions_i32), p(reward), p(next_obs), p(done_u8), p(self.states),
                                          p(self.next_states), p(self.actions), p(self.rewards), p(self.dones),
                                          n, self.ptr, self.capacity, stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_replay_append failed ({rc})")
        m = min(n, self.capacity)
        self.ptr = (self.ptr + m) % self.capacity
        self.size = min(self.capacity, self.size + m)
        self.version += 1

    def add_batch(self, states, actions, rewards, next_states, dones):
        """n transitions at once; FIFO eviction like deque(maxlen).  With n_step > 1 row i is the next transition of stream i."""
        if self.n_step > 1:
            out = self._nstep_window(states, actions, rewards, next_states, dones)
            if out is None:
                return
            states, actions, rewards, next_states, dones = out
        n = states.shape[0]
        if n > self.capacity:   # only the newest `capacity` survive
            states, actions, rewards, next_states, dones = (t[-self.capacity:] for t in (states, actions, rewards, next_states, dones))
            n = self.capacity
        end = self.ptr + n
        if end <= self.capacity:
            sl = slice(self.ptr, end)
            self.states[sl].copy_(states); self.next_states[sl].copy_(next_states)
            self.actions[sl, 0].copy_(actions.view(-1)); self.rewards[sl, 0].copy_(rewards.view(-1))
            self.dones[sl, 0].copy_(dones.view(-1))
        else:
            k = self.capacity - self.ptr
            n_step, self.n_step = self.n_step, 1      # the two halves are already n-step transitions
            try:
                self.add_batch(states[:k], actions[:k], rewards[:k], next_states[:k], dones[:k])
                self.add_batch(states[k:], actions[k:], rewards[k:], next_states[k:], dones[k:])
            finally:
                self.n_step = n_step
            return
        self.ptr = end % self.capacity
        self.size = min(self.capacity, self.size + n)
        self.version += 1

    def advance(self, n):
        """Ring bookkeeping after `n` transitions were written at `ptr` by a kernel (mn_step_append)."""
        m = min(int(n), self.capacity)
        self.ptr = (self.ptr + m) % self.capacity
        self.size = min(self.capacity, self.size + m)
        self.version += 1

    def sample_indices(self, b):
        """`b` distinct uniform row indices in [0, size) (random.sample, replay_buffer.py:47) without a full
        permutation of the ring: draw with replacement, keep first occurrences, top up until b are distinct (each
        accepted index is uniform over what the earlier ones left, i.e. sequential sampling without replacement)."""
        assert self.size >= b
        if self.size <= 4 * b:      # dense case: a permutation is the cheap way
            return torch.randperm(self.size, device=self.device, generator=self.gen)[:b]
        chosen = torch.empty(0, dtype=torch.int64, device=self.device)
        while chosen.numel() < b:
            cand = torch.randint(0, self.size, (2 * b,), device=self.device, generator=self.gen)
            allv = torch.cat([chosen, cand])
            # stable first-occurrence filter
            srt, order = torch.sort(allv, stable=True)
            first = torch.ones_like(srt, dtype=torch.bool)
            first[1:] = srt[1:] != srt[:-1]
            keep = torch.zeros_like(first)
            keep[order] = first
            chosen = allv[keep][:b]
        return chosen

    def sample(self, batch_size=None):
        """Uniform without replacement (random.sample, replay_buffer.py:47) -> float32 / int64 tensors."""
        b = self.batch_size if batch_size is None else batch_size
        idx = self.sample_indices(b)
        return (self.states[idx], self.actions[idx], self.rewards[idx], self.next_states[idx], self.dones[idx])

    def __len__(self):
        return self.size
