"""IQN agent: batched act / learn around the HIP vector env.

Keeps the reference's `IQNAgent` surface (thirdparty/IQN/agent.py:10-407): the constructor keywords, `train`,
`soft_update`, `load_model` here; the reference-shaped single-env methods (`learn`, `evaluation`, `act`, `act_eval`,
`act_adaptive`, `adjust_cvar`, `linear_eps`) are the boundary layer and live in `iqn/compat.py` (`ReferenceLoopMixin`).
This file is the batched code the MI355X path actually runs: `act_batch`, `vec_step`, `learn_vec`, `evaluation_vec`,
the fused / pipelined gradient step.  Optional data-parallel training over RCCL:
one flat 35 785-float gradient bucket, one all_reduce per grad step.
"""
import json
import os
import random
import sys

import numpy as np
import torch
import torch.optim as optim

from .cadence import cadence_tick
from .compat import ReferenceLoopMixin
from .model import ObsEncoder
from .replay_buffer import ReplayBuffer


def calculate_huber_loss(td_errors, k=1.0):
    """agent.py:401-407, element-wise Huber with threshold k."""
    return torch.where(td_errors.abs() <= k, 0.5 * td_errors.pow(2), k * (td_errors.abs() - 0.5 * k))


class UnderActGuard:
    """Keeps `IQNAgent.reset_under_act` honest on the box the loop actually runs on.

    Episode resets under the next act kernel (mn_reset_done_async) need the reset launch to run BESIDE that kernel: a hardware queue of its own and room
    on the CUs next to the act workgroups.  HIP promises neither (another ROCm release, a partitioned or shared GPU).  When it does not happen, every late
    row waits out its bound and is then taken as it is.  So the first `preflight` vector steps of a loop run with EVERY reset forced under the act kernel
    (whatever the library's episode-count rule would decide), the late-row waits that ran out are read after step 2 and after the last preflight step,
    and -- if there were any -- the loop goes back to resets in front of the act kernel with one line on stderr instead of failing at its first evaluation
    point.  After that the count is looked at every `poll_every` steps without synchronising (mn_iqn_late_timeouts_peek).  Call `after_step(i)` behind
    vector step i = 0, 1, ...; `close()` restores the library's rule if the loop ends inside the preflight.
    The preflight steps ARE steps of the run: with the resets served in time their results are those of resets in front, bit for bit; if not, a dozen
    vector steps at the start of a run chose some actions from unfinished observation rows (at eps ~ 1, where the choice is random anyway)."""

    def __init__(self, agent, env, preflight=12, poll_every=64, log=None):
        self.agent, self.env = agent, env
        self.preflight, self.poll_every = int(preflight), int(poll_every)
        self.log = log
        self.fallback = None      # dict(step=, timeouts=) once the loop has gone back to resets in front
        self.active = bool(agent.reset_under_act) and agent.device.type == "cuda" and agent.use_fused_act and hasattr(env, "set_reset_under_act_max")
        self._restore = None
        if self.active:
            self._seen0 = agent._late_acknowledged      # (time-outs an earlier loop of this agent has already answered with its fallback do not count again)
            if self.preflight > 0:
                self._restore = env.reset_under_act_max
                env.set_reset_under_act_max(2 ** 31 - 1)

    def _end_preflight(self):
        if self._restore is not None:
            self.env.set_reset_under_act_max(self._restore)
            self._restore = None

    def _fall_back(self, step, k):
        self.agent.reset_under_act = False
        self.env.join_reset()
        self._end_preflight()
        self.fallback = dict(step=int(step), timeouts=int(k))
        self.agent._late_acknowledged = self._seen0 + int(k)
        self.active = False
        msg = (f"reset_under_act: {k} act rows were taken before their episode reset had finished (by vector step {step}): the reset launch does not run beside the act "
               "kernel on this device -- episode resets go in front of the act kernel from here on (mn_reset_done)")
        (self.log or (lambda m: print(m, file=sys.stderr, flush=True)))(msg)

    def after_step(self, i):
        if not self.active:
            return
        from .fused_act import late_timeouts, late_timeouts_peek
        if i < self.preflight:
            if i == 1 or i == self.preflight - 1:
                self.env.join_reset()
                k = late_timeouts(self.agent.qnetwork_local) - self._seen0      # (synchronises: twice per loop)
                if k > 0:
                    return self._fall_back(i, k)
                if i == self.preflight - 1:
                    self._end_preflight()
        elif self.poll_every > 0 and (i - self.preflight) % self.poll_every == self.poll_every - 1:
            k = late_timeouts_peek(self.agent.qnetwork_local) - self._seen0
            if k > 0:
                self._fall_back(i, k)

    def close(self):
        self._end_preflight()


class IQNAgent(ReferenceLoopMixin):
    def __init__(self, state_size, action_size, layer_size=64, n_step=1, BATCH_SIZE=32, BUFFER_SIZE=1_000_000,
                 LR=1e-4, TAU=1.0, GAMMA=0.99, UPDATE_EVERY=4, learning_starts=10000, target_update_interval=10000,
                 exploration_fraction=0.1, initial_eps=1.0, final_eps=0.05, device="cpu", seed=0,
                 distributed=False, act_chunk=8192, rank=0):
        self.state_size = state_size
        self.action_size = action_size
        self.device = torch.device(device)
        self.LR, self.TAU, self.GAMMA = LR, TAU, GAMMA
        self.UPDATE_EVERY = UPDATE_EVERY
        self.BATCH_SIZE = BATCH_SIZE
        self.n_step = n_step
        self.learning_starts = learning_starts
        self.target_update_interval = target_update_interval
        self.exploration_fraction = exploration_fraction
        self.initial_eps, self.final_eps = initial_eps, final_eps
        self.N = 8                                   # train-time quantile samples (agent.py:286,290)
        self.rank = int(rank)                        # shared learner: same `seed` (identical init) on every rank, but
                                                     # rank-specific exploration / tau / replay-sampling streams
        self.act_chunk = act_chunk
        self.grad_steps_per_update = 1               # vectorised loop only: grad steps per training event
        self.use_fused_act = True                    # GPU tensors: fused HIP act kernel (csrc/iqn_act.hip)
        self._act_rng = None                         # the act path's own counter-based tau / exploration draws (fused_act.ActRng)
        self.use_library_rng = True                  # False: taus / exploration uniforms from torch.rand on self.gen
        self.shared_taus = False                     # opt-in: one set of 32 taus per act LAUNCH instead of per row (fused_act(shared_taus=True))
        self.use_fused_graph = False                 # opt-in: the fused gradient steps of one training event as one captured hipGraph (train_steps_from_memory)
        self.reset_under_act = False                 # vec_step: the episode resets of a vector step run on the env's own stream UNDER the next step's act kernel (which takes
                                                     # the finished envs' rows last) instead of in front of it: same results, ~28 us off every vector step's critical path.  The
                                                     # `obs` vec_step returns then has rows still being written: hand it back to vec_step, or call `train_env.join_reset()` before
                                                     # reading it.  On in learn_vec / train_iqn / bench.py; off by default for callers that look at `obs` between steps
        self.use_train_graph = False                 # opt-in: grad step replayed from a captured hipGraph (measured: no gain, the step is bound by kernel time, not launches)
        self._late_acknowledged = 0                  # late-row time-outs of the act context that a loop has already answered by going back to resets in front (UnderActGuard)
        self.under_act_fallback = None               # learn_vec: dict(step, timeouts) if that happened in the last loop
        self._graph = None
        self._graph_bypass_logged = False
        # GPU: the whole optimizer step as five HIP kernels (csrc/iqn_train.hip: sample, forward+backward, reduce, norm,
        # Adam) instead of ~150 PyTorch autograd / Adam kernels; False = the PyTorch path (always used on the CPU)
        self.use_fused_train = torch.device(device).type == "cuda"
        self._fused = None

        self.qnetwork_local = ObsEncoder(state_size, action_size, seed, device)
        self.qnetwork_target = ObsEncoder(state_size, action_size, seed, device)   # identical init (App. A A1)
        self.optimizer = self._make_adam()
        self.memory = ReplayBuffer(BUFFER_SIZE, BATCH_SIZE, device, seed, GAMMA, n_step, state_size)
        random.seed(seed)                            # replay_buffer.py:21 seeds python `random` (eps-greedy)
        self.gen = torch.Generator(device=self.device)
        self.gen.manual_seed(int(seed) + 12345 + 7919 * self.rank)
        if self.rank:
            self.memory.gen.manual_seed(int(seed) + 104729 * self.rank)
        self.target_sync_grad_steps = None           # vectorised loop: hard target copy every this many grad steps
                                                     # (None: every target_update_interval vector steps, see vec_step)
        self._last_sync_at = 0
        self._train_path = None                      # "hip" / "torch": which gradient step ran last (Adam step-count hand-over)

        self.current_timestep = 0
        self.learning_timestep = 0
        self.grad_steps = 0
        self.distributed = bool(distributed)
        self.exchange = "collective"                 # shared learner: "collective" = RCCL all-reduce of the flat gradient (default); "mailbox" =
                                                     # one-shot exchange over IPC-mapped mailboxes (iqn/mailbox.py; opt-in)
        self._flat = None

        self.eval_timesteps = dict(greedy=[], adaptive=[])
        self.eval_actions = dict(greedy=[], adaptive=[])
        self.eval_rewards = dict(greedy=[], adaptive=[])
        self.eval_successes = dict(greedy=[], adaptive=[])
        self.eval_times = dict(greedy=[], adaptive=[])
        self.eval_energies = dict(greedy=[], adaptive=[])
        self.best_eval = None                        # learn_vec: the best greedy evaluation so far (successes, mean return) and when

    def _make_adam(self):
        """Adam(lr=1e-4) as agent.py:66; on the GPU the fused single-kernel implementation (same update rule)."""
        fused = torch.device(self.device).type == "cuda" and os.environ.get("MN_FUSED_ADAM", "1") == "1"
        return optim.Adam(self.qnetwork_local.parameters(), lr=self.LR, fused=fused) if fused else \
            optim.Adam(self.qnetwork_local.parameters(), lr=self.LR)

    # ---- checkpoints ---------------------------------------------------------------------------
    def load_model(self, path, device="cpu"):
        """agent.py:86-92."""
        self.qnetwork_local = ObsEncoder.load(path, device)
        self.qnetwork_target = ObsEncoder.load(path, device)
        self.device = torch.device(device)
        self.optimizer = self._make_adam()
        self._train_path = None

    def adjust_cvar_batch(self, states):
        """Batched adjust_cvar on device: states [n, 26] -> cvar [n]."""
        p = states[:, 4:].view(states.shape[0], -1, 2)
        skip = (p[:, :, 0].abs() < 1e-3) & (p[:, :, 1].abs() < 1e-3)
        d = torch.linalg.vector_norm(p, dim=2).masked_fill(skip, float("inf"))
        closest = d.min(dim=1).values
        return torch.where(closest < 10.0, closest / 10.0, torch.ones_like(closest))

    # ---- acting --------------------------------------------------------------------------------
    @torch.no_grad()
    def act_eval_batch(self, states, eps=0.0, cvar=1.0, taus=None):
        """Batched act_eval (agent.py:217-236): states [n,26] on the device -> (actions [n] i32, quantiles [n,32,9],
        taus [n,32,1]) -- the per-action return distribution samples and the (cvar-scaled) quantile fractions they
        were evaluated at, as run_experiments.py:26-69 records them.  `cvar` is a float or a per-row tensor."""
        if states.is_cuda and self.use_fused_act:
            from .fused_act import fused_act, ActRng
            if taus is None and self.use_library_rng:
                if self._act_rng is None:
                    self._act_rng = ActRng(self.gen.initial_seed(), states.device)
                return fused_act(self.qnetwork_local, states.contiguous(), eps, cvar, rng=self._act_rng, want_quantiles=True,
                                 shared_taus=self.shared_taus)
            return fused_act(self.qnetwork_local, states.contiguous(), eps, cvar, taus=taus, generator=self.gen,
                             want_quantiles=True, shared_taus=self.shared_taus and (taus is None or taus.numel() == self.qnetwork_local.K))
        quantiles, t = self.qnetwork_local.forward(states, self.qnetwork_local.K, cvar, taus=taus)
        greedy = quantiles.mean(dim=1).argmax(dim=1).to(torch.int32)
        if eps > 0.0:
            n = states.shape[0]
            u = torch.rand(n, device=states.device, generator=self.gen)
            rnd = torch.randint(0, self.action_size, (n,), device=states.device, dtype=torch.int32, generator=self.gen)
            greedy = torch.where(u > eps, greedy, rnd)
        return greedy, quantiles, t

    @torch.no_grad()
    def qvals_batch(self, states, cvar=1.0, taus=None):
        """Q(s, .) = mean over K = 32 quantile samples, for a whole vector of states (device tensor).
        On the GPU this is the fused HIP kernel (csrc/iqn_act.hip); on CPU tensors (tests) plain PyTorch,
        chunked over envs."""
        if states.is_cuda and self.use_fused_act:
            from .fused_act import fused_qvals
            return fused_qvals(self.qnetwork_local, states, cvar, taus=taus, generator=self.gen)
        n = states.shape[0]
        out = torch.empty(n, self.action_size, dtype=torch.float32, device=states.device)
        step = self.act_chunk
        for lo in range(0, n, step):
            hi = min(n, lo + step)
            c = cvar[lo:hi] if torch.is_tensor(cvar) else cvar
            t = taus[lo:hi] if taus is not None else None
            out[lo:hi] = self.qnetwork_local.get_qvals(states[lo:hi], c, taus=t)
        return out

    @torch.no_grad()
    def act_batch(self, states, eps, cvar=1.0, late_env=None):
        """Batched eps-greedy act (agent.py:186-205 per row): states [n,26] f32 on device ->
        actions [n] int32 on device.  On the GPU this is ONE fused HIP kernel (network, mean over taus,
        argmax, exploration); exploration draws come from a device generator.
        `late_env`: the env whose `reset_done(under_next_act=True)` may still be writing rows of `states` (vec_step with `reset_under_act`):
        the fused kernel takes those rows last; every other path waits for the reset first."""
        if states.is_cuda and self.use_fused_act:
            from .fused_act import fused_act
            if not self.use_library_rng:
                return fused_act(self.qnetwork_local, states.contiguous(), eps, cvar, generator=self.gen, shared_taus=self.shared_taus, late_env=late_env)
            if self._act_rng is None:
                from .fused_act import ActRng
                self._act_rng = ActRng(self.gen.initial_seed(), states.device)
            return fused_act(self.qnetwork_local, states.contiguous(), eps, cvar, rng=self._act_rng, shared_taus=self.shared_taus, late_env=late_env)
        if late_env is not None:
            late_env.join_reset()
        q = self.qvals_batch(states, cvar)
        greedy = q.argmax(dim=1).to(torch.int32)
        if eps <= 0.0:
            return greedy
        n = states.shape[0]
        u = torch.rand(n, device=states.device, generator=self.gen)
        rnd = torch.randint(0, self.action_size, (n,), device=states.device, dtype=torch.int32, generator=self.gen)
        return torch.where(u > eps, greedy, rnd)

    # ---- learning ------------------------------------------------------------------------------
    def _allreduce_grads(self):
        import torch.distributed as dist
        params = [p for p in self.qnetwork_local.parameters() if p.grad is not None]
        grads = [p.grad for p in params]
        if self._flat is None or self._flat.numel() != sum(g.numel() for g in grads):
            self._flat = torch.empty(sum(g.numel() for g in grads), dtype=torch.float32, device=grads[0].device)
        torch.cat([g.reshape(-1) for g in grads], out=self._flat)
        dist.all_reduce(self._flat, op=dist.ReduceOp.SUM)       # one 143 KB bucket over RCCL/xGMI
        self._flat.div_(dist.get_world_size())
        off = 0
        for g in grads:
            g.copy_(self._flat[off:off + g.numel()].view_as(g))
            off += g.numel()

    def compute_loss(self, experiences, taus_target=None, taus_local=None):
        """Quantile-Huber TD loss of agent.py:276-295 (taus can be injected for tests)."""
        states, actions, rewards, next_states, dones = experiences
        B = states.shape[0]
        with torch.no_grad():
            Q_targets_next, _ = self.qnetwork_target(next_states, self.N, taus=taus_target)
            Q_targets_next = Q_targets_next.max(2)[0].unsqueeze(1)                          # (B, 1, N)
            Q_targets = rewards.unsqueeze(-1) + (self.GAMMA ** self.n_step * Q_targets_next * (1. - dones.unsqueeze(-1)))
        Q_expected, taus = self.qnetwork_local(states, self.N, taus=taus_local)
        Q_expected = Q_expected.gather(2, actions.unsqueeze(-1).expand(B, self.N, 1))
        # td_error[b, i, j] = Q_targets[b, 0, j] - Q_expected[b, i, 0]   (B, N, N); Huber with kappa = 1
        # (agent.py:401-407) as ONE fused op with its own backward instead of abs / le / pow / where chains
        qt, qe = Q_targets.expand(B, self.N, self.N), Q_expected.expand(B, self.N, self.N)
        huber_l = torch.nn.functional.huber_loss(qe, qt, reduction="none", delta=1.0)
        with torch.no_grad():
            weight = (taus - (qt < qe).to(taus.dtype)).abs()      # |tau - 1[td < 0]|, agent.py:293
        # sum over the local-quantile axis, mean over the target-sample axis, mean over the batch (agent.py:294-295)
        return (weight * huber_l).sum() / (B * self.N)

    def _fused_trainer(self):
        from .fused_train import FusedTrainer
        if self._fused is None or not self._fused.owns(self):      # load_model() replaces the networks
            self._fused = FusedTrainer(self)
        return self._fused

    def _enter_train_path(self, path):
        """Both gradient-step paths update ONE Adam state (the moments are shared memory, iqn/fused_train.py); the step
        count is handed over whenever the path changes, so flipping `use_fused_train` mid-run continues the same
        optimizer instead of restarting its bias correction."""
        if self._train_path == path:
            return
        if self._fused is not None and self._fused.owns(self):
            if path == "torch" and self._train_path == "hip":
                self._fused.sync_to_optimizer(self.optimizer)
            elif path == "hip" and self._train_path == "torch":
                self._fused.sync_from_optimizer(self.optimizer)
        self._train_path = path

    def train_steps_from_memory(self, n_steps):
        """`n_steps` x train_from_memory().  With `use_fused_graph` (GPU, fused gradient step) the whole sequence -- every step's
        forward / backward, reduction, RCCL all-reduce of a shared learner, Adam -- is ONE hipGraph launch (iqn/fused_train.py:
        graphed_steps): the host enqueues one node instead of 3-4 launches per step, which is what a shared learner's 16 steps per
        vector step need to stay ahead of the GPU.  Same arithmetic, same generator stream: bit-identical to the eager calls."""
        # (while the ring is still filling its row count changes with every vector step and each change would be a re-capture +
        # device synchronisation: the eager steps -- bit-identical -- run until the ring is full)
        want_graph = self.use_fused_graph and self.use_fused_train and self.device.type == "cuda" and n_steps > 1
        if want_graph and len(self.memory) < self.memory.capacity and not self._graph_bypass_logged:
            self._graph_bypass_logged = True
            print(f"[IQNAgent] use_fused_graph: eager gradient steps until the replay ring is full ({len(self.memory)} of {self.memory.capacity} rows)", flush=True)
        if want_graph and len(self.memory) >= self.BATCH_SIZE and len(self.memory) == self.memory.capacity:
            m = self.memory
            ft = self._fused_trainer()
            self._enter_train_path("hip")
            loss = ft.graphed_steps((m.states, m.actions, m.rewards, m.next_states, m.dones), m.size, self.BATCH_SIZE, n_steps)
            self.grad_steps += n_steps
            return loss
        loss = None
        for _ in range(n_steps):
            loss = self.train_from_memory()
        return loss

    def train_from_memory(self):
        """`self.train(self.memory.sample())` (agent.py:131-133).  With `use_fused_train` the HIP step gathers its batch
        straight from the device ring (no sampled copies)."""
        if self.use_fused_train and self.device.type == "cuda":
            m = self.memory
            ft = self._fused_trainer()
            self._enter_train_path("hip")
            # replay_buffer.py:47 + model.py:149 + agent.py:269-304: the batch is drawn inside the forward / backward launch
            loss = ft.step_sampled((m.states, m.actions, m.rewards, m.next_states, m.dones), m.size, self.BATCH_SIZE, m.version)
            self.grad_steps += 1
            return loss
        return self.train(self.memory.sample())

    def train(self, experiences, taus_target=None, taus_local=None):
        """agent.py:269-304: one optimizer step; returns the loss (device scalar tensor).
        GPU tensors with `use_fused_train` (the default on the GPU): the hand-written HIP step (csrc/iqn_train.hip:
        both forwards, quantile-Huber loss, backward, clip, Adam in four launches).  Otherwise PyTorch autograd +
        torch.optim.Adam -- the definition the HIP step is tested against and the only path on the CPU; with
        `use_train_graph` (opt-in, single learner, taus not injected) replayed from one captured hipGraph."""
        if self.use_fused_train and experiences[0].is_cuda:
            exp = tuple(t.contiguous() for t in experiences)
            ft = self._fused_trainer()
            self._enter_train_path("hip")
            loss = ft.step(exp, None, taus_target, taus_local)
            self.grad_steps += 1
            return loss
        self._enter_train_path("torch")
        if (self.use_train_graph and experiences[0].is_cuda and not self.distributed
                and taus_target is None and taus_local is None and experiences[0].shape[0] == self.BATCH_SIZE):
            return self._train_graphed(experiences)
        self.optimizer.zero_grad(set_to_none=False)
        loss = self.compute_loss(experiences, taus_target, taus_local)
        loss.backward()
        if self.distributed:
            self._allreduce_grads()          # average first, then clip: same as one big-batch learner
        torch.nn.utils.clip_grad_norm_(self.qnetwork_local.parameters(), 0.5)
        self.optimizer.step()
        self.grad_steps += 1
        return loss.detach()

    def _train_graphed(self, experiences):
        if self._graph is None:
            dev = experiences[0].device
            self._g_in = tuple(torch.empty_like(t) for t in experiences)
            for dst, src in zip(self._g_in, experiences):
                dst.copy_(src)
            # Adam must keep its step counter on the device to be capturable; same arithmetic
            state = self.optimizer.state_dict()
            self.optimizer = optim.Adam(self.qnetwork_local.parameters(), lr=self.LR, capturable=True)
            if state["state"]:
                for st in state["state"].values():
                    st["step"] = torch.as_tensor(st["step"], dtype=torch.float32, device=dev)
                self.optimizer.load_state_dict(state)
                for g_ in self.optimizer.param_groups:
                    g_["capturable"] = True
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream(dev))
            snap = [p.detach().clone() for p in self.qnetwork_local.parameters()]
            # the optimizer's moment tensors may be views of the fused trainer's flat buffers (one Adam state for both
            # paths, iqn/fused_train.py) and load_state_dict keeps the tensors it is given: snapshot the VALUES, the
            # warm-up steps below update them in place
            import copy
            opt_snap = {k: {kk: (vv.detach().clone() if torch.is_tensor(vv) else copy.deepcopy(vv)) for kk, vv in st.items()}
                        for k, st in state["state"].items()}
            with torch.cuda.stream(side):   # warm-up iterations on a side stream (allocations, lazy init)
                for _ in range(3):
                    self.optimizer.zero_grad(set_to_none=True)
                    self.compute_loss(self._g_in).backward()
                    torch.nn.utils.clip_grad_norm_(self.qnetwork_local.parameters(), 0.5)
                    self.optimizer.step()
            torch.cuda.current_stream(dev).wait_stream(side)
            # the warm-up must not count as training: restore weights and optimizer state (values copied back INTO the
            # tensors the optimizer holds, so moments shared with the fused trainer stay shared)
            with torch.no_grad():
                for p_, s_ in zip(self.qnetwork_local.parameters(), snap):
                    p_.copy_(s_)
                live = self.optimizer.state_dict()["state"]
                for k, st in opt_snap.items():
                    for kk, vv in st.items():
                        if torch.is_tensor(vv):
                            live[k][kk].copy_(vv)
                if not opt_snap:      # the optimizer had no state before the warm-up created it: zero what it made
                    for st in live.values():
                        for vv in st.values():
                            if torch.is_tensor(vv):
                                vv.zero_()
            if self._fused is not None and self._fused.owns(self):
                self._fused._adopt_optimizer_state(self.optimizer)     # re-attach the flat moment buffers to the new optimizer
            self._graph = torch.cuda.CUDAGraph()
            self.optimizer.zero_grad(set_to_none=True)
            with torch.cuda.graph(self._graph):
                loss = self.compute_loss(self._g_in)
                loss.backward()
                torch.nn.utils.clip_grad_norm_(self.qnetwork_local.parameters(), 0.5)
                self.optimizer.step()
                self._g_loss = loss.detach()
            # capture does not execute: weights / optimizer untouched so far
        for dst, src in zip(self._g_in, experiences):
            dst.copy_(src)
        self._graph.replay()
        # a graph replay writes the weights without bumping the parameters' version counters: the act path's cached
        # weight image (iqn/fused_act.py) has to be told (writes through `.data` / raw pointers need the same call)
        from .fused_act import weights_changed
        weights_changed(self.qnetwork_local)
        self.grad_steps += 1
        return self._g_loss

    def soft_update(self, local_model, target_model):
        """agent.py:307-317 (TAU = 1.0 -> hard copy)."""
        with torch.no_grad():
            ft = self._fused
            if (ft is not None and ft.owns(self) and local_model is self.qnetwork_local
                    and target_model is self.qnetwork_target):
                # both networks are views of two flat buffers (iqn/fused_train.py): one copy instead of 14
                if self.TAU == 1.0:
                    ft.target.copy_(ft.local)
                else:
                    ft.target.mul_(1.0 - self.TAU).add_(ft.local, alpha=self.TAU)
                return
            for tp, lp in zip(target_model.parameters(), local_model.parameters()):
                tp.data.copy_(self.TAU * lp.data + (1.0 - self.TAU) * tp.data)

    def _sync_target(self):
        """Target copy + the gradient-step mark the `target_sync_grad_steps` cadence counts from (iqn/cadence.py)."""
        self.soft_update(self.qnetwork_local, self.qnetwork_target)
        self._last_sync_at = self.grad_steps

    # ---- batched loop on the HIP vector env ----------------------------------------------------------
    def learn_vec(self, total_vector_steps, train_env, eval_env=None, eval_config=None, eval_freq=None,
                  eval_log_path=None, total_timesteps=None, world_size=1, cvar=1.0, verbose=True,
                  train_every=None, on_step=None, report_timestep_scale=1.0, eval_adaptive=True, reset_under_act=True):
        """Vectorised agent.py:94-173.  One iteration = one vector step of `train_env` (n_envs env
        steps): act_batch -> mn_step -> replay.add_batch -> mn_reset_done -> (every UPDATE_EVERY vector
        steps) sample + train.  `current_timestep` counts env steps over all ranks, so eps, the
        learning_starts gate and the curriculum keep the reference's meaning of "timesteps";
        `learning_timestep` counts vector steps after learning_starts (UPDATE_EVERY,
        target_update_interval and eval_freq are applied to it)."""
        n = train_env.n_envs
        per_iter = n * world_size
        # evaluation npz `timesteps` are reported as current_timestep * report_timestep_scale (train_iqn: reference-
        # equivalent timesteps, so scripts/plot_eval_returns.py keeps its x axis)
        self._report_scale = float(report_timestep_scale)
        if total_timesteps is None:
            total_timesteps = total_vector_steps * per_iter
        train_every = self.UPDATE_EVERY if train_every is None else train_every
        obs = train_env.reset()
        ep_ret = torch.zeros(n, device=self.device)
        ep_len = torch.zeros(n, device=self.device)
        stats = dict(episodes=0, successes=0, collisions=0, timeouts=0, loss=None)
        # this loop never looks at `obs` between two vector steps, so the resets can run under the next step's act kernel (see `reset_under_act`) -- checked on this
        # device by the loop's first steps (UnderActGuard: falls back to resets in front with one log line if the reset launch does not run beside the act kernel)
        was_under_act, self.reset_under_act = self.reset_under_act, bool(reset_under_act)
        guard = UnderActGuard(self, train_env)
        self.under_act_fallback = None
        try:
            for it in range(total_vector_steps):
                eps = self.linear_eps(total_timesteps)
                evaluate_now = eval_env is not None and cadence_tick(self, train_every, eval_freq).evaluate      # (the state vec_step's own tick sees)
                obs, reward, done, info, loss = self.vec_step(train_env, obs, eps, cvar, train_every, per_iter)
                guard.after_step(it)
                if loss is not None:
                    stats["loss"] = loss
                if verbose:
                    ep_ret += (train_env.discount ** ep_len) * reward
                    ep_len += 1
                    d = done.bool()
                    stats["episodes"] += int(d.sum())
                    stats["successes"] += int((info == 4).sum())
                    stats["collisions"] += int((info == 3).sum())
                    stats["timeouts"] += int((info == 2).sum())
                    ep_ret.masked_fill_(d, 0.0); ep_len.masked_fill_(d, 0.0)
                if evaluate_now:
                    self.check_learner()      # (a device synchronisation; the evaluation below is one anyway)
                    res = self.evaluation_vec(eval_env, eval_config, greedy=True, eval_log_path=eval_log_path)
                    if eval_adaptive:
                        self.evaluation_vec(eval_env, eval_config, greedy=False, eval_log_path=eval_log_path)
                    # agent.py:140-148 keeps the LATEST network at every evaluation point; the batched run also keeps the BEST greedy evaluation so far beside it
                    # (`best_*`: ~1 run in 12 ends on a checkpoint far below its own best -- profiles/r05_learning_curve.txt)
                    score = (int(sum(res["successes"])), float(np.mean(res["rewards"])))
                    if self.best_eval is None or score > self.best_eval["score"]:
                        self.best_eval = dict(score=score, timestep=self.eval_timesteps["greedy"][-1], grad_steps=self.grad_steps, vector_step=it)
                        if eval_log_path is not None:
                            self.qnetwork_local.save(eval_log_path, prefix="best_")
                            with open(os.path.join(eval_log_path, "best_evaluation.json"), "w") as f:
                                json.dump(dict(successes=score[0], n_worlds=len(res["successes"]), mean_return=score[1], **{k: v for k, v in self.best_eval.items() if k != "score"}), f)
                    if eval_log_path is not None:
                        self.qnetwork_local.save(eval_log_path)
                if on_step is not None:
                    on_step(it, stats)
            # the end-of-run look at the bounded waits happens while `reset_under_act` still says how this loop ran (every rank, evaluation env or not)
            if hasattr(train_env, "join_reset"):
                train_env.join_reset()
            self.check_learner()
        finally:
            guard.close()
            self.under_act_fallback = guard.fallback
            self.reset_under_act = was_under_act
            if hasattr(train_env, "join_reset"):
                train_env.join_reset()
        return stats

    def check_learner(self):
        """Raise if a bounded wait of the fused gradient step ran out since the last look (`FusedTrainer.check_timeouts`): the step(s) concerned updated
        nothing, and with the mailbox exchange a peer's gradient did not arrive -- the ranks of a shared learner would drift apart silently otherwise."""
        if self._fused is not None and self._fused.owns(self):
            self._fused.check_timeouts()
        if self.reset_under_act and self.device.type == "cuda" and self.use_fused_act:
            from .fused_act import late_timeouts
            k = late_timeouts(self.qnetwork_local) - self._late_acknowledged
            if k > 0:
                raise RuntimeError(f"{k} act rows were taken before their episode reset had finished (mn_iqn_late_timeouts): the reset launch did not run beside the act kernel")

    def vec_step(self, train_env, obs, eps, cvar=1.0, train_every=None, per_iter=None):
        """One iteration of the vectorised loop: act_batch -> mn_step -> replay.add_batch ->
        mn_reset_done -> (cadence permitting) sample + train + target sync.  Everything is enqueued on
        the current HIP stream; nothing synchronises with the host.
        Returns (obs for the next act, reward, done, info, loss or None)."""
        train_every = self.UPDATE_EVERY if train_every is None else train_every
        per_iter = train_env.n_envs if per_iter is None else per_iter
        under_act = bool(self.reset_under_act) and obs.is_cuda and hasattr(train_env, "take_late_rows") and self.use_fused_act
        if under_act:      # only the default acting form takes late rows: with any other the reset stays in front (no cross-stream events for nothing)
            from .fused_act import late_rows_possible
            under_act = late_rows_possible(self.qnetwork_local, obs.shape[0], self.shared_taus)
        actions = self.act_batch(obs, eps, cvar, late_env=train_env if under_act or getattr(train_env, "late_rows", None) is not None else None)
        if obs.is_cuda and hasattr(train_env, "step_append") and self.n_step == 1:
            # mn_step_append: the step kernel itself writes (obs_t, a, r, obs_t+1 incl. terminal observations, done)
            # into the replay ring -- no separate append launch, obs_t+1 is not re-read
            next_obs, reward, done, info = train_env.step_append(actions, obs, self.memory)
        else:
            next_obs, reward, done, info = train_env.step(actions)      # other half of the double buffer
            if obs.is_cuda:   # terminal obs, appended before the reset overwrites the finished rows
                self.memory.add_vector_step(obs, actions, reward, next_obs, done)
            else:
                self.memory.add_batch(obs, actions, reward, next_obs, done.float())
        loss = None
        due = cadence_tick(self, train_every)      # iqn/cadence.py: agent.py:126-147's rule (+ the target cadence in gradient steps)
        # first observations where done (`reset_under_act`: on the env's own stream, under the next call's act kernel, which takes those rows last -- launched BEHIND the
        # training event: a reset wavefront finds room beside an act workgroup, not beside a gradient step's, whose launches it would only hold up)
        if not under_act:
            obs = train_env.reset_done()
        if due.train:
            loss = self.train_steps_from_memory(self.grad_steps_per_update)      # 1 = the reference's cadence (agent.py:129-133)
        if due.sync:
            self._sync_target()
        if under_act:
            obs = train_env.reset_done(under_next_act=True)
        if self.current_timestep >= self.learning_starts:
            self.learning_timestep += 1
        self.current_timestep += per_iter
        return obs, reward, done, info, loss

    @torch.no_grad()
    def evaluation_vec(self, eval_env, eval_config, greedy=True, eval_log_path=None, max_steps=1000):
        """agent.py:319-398 with all evaluation worlds stepped side by side on the GPU.
        `eval_env` is a VecMarineNavEnv with n_envs == len(eval_config); the npz schema is unchanged."""
        from ..marinenav_env.vec_env import VecMarineNavEnv
        cfgs = list(eval_config.values())
        n = len(cfgs)
        assert eval_env.n_envs == n
        r0 = cfgs[0]["robot"]
        eval_env.set_attrs(N=r0["N"], dt=r0["dt"])
        obs = eval_env.load_worlds([VecMarineNavEnv.world_from_eval_config(c) for c in cfgs]).clone()
        a_tab = torch.tensor(r0["a"], device=self.device); w_tab = torch.tensor(r0["w"], device=self.device)
        e_a = (a_tab / a_tab.max()).abs(); e_w = (w_tab / w_tab.max()).abs()
        energy_tab = (e_a.view(3, 1) + e_w.view(1, 3)).reshape(-1)      # robot.py:72-77
        alive = torch.ones(n, dtype=torch.bool, device=self.device)
        ret = torch.zeros(n, dtype=torch.float64, device=self.device)
        length = torch.zeros(n, dtype=torch.int64, device=self.device)
        energy = torch.zeros(n, dtype=torch.float64, device=self.device)
        last_info = torch.zeros(n, dtype=torch.uint8, device=self.device)
        acts = torch.full((max_steps, n), -1, dtype=torch.int32, device=self.device)
        self.qnetwork_local.eval()
        for t in range(max_steps):
            cv = 1.0 if greedy else self.adjust_cvar_batch(obs)
            a = self.act_batch(obs, 0.0, cv)
            obs, reward, done, info = eval_env.step(a)
            ret += torch.where(alive, (eval_env.discount ** t) * reward.double(), torch.zeros_like(ret))
            length += alive.long()
            energy += torch.where(alive, energy_tab[a.long()].double(), torch.zeros_like(energy))
            acts[t] = torch.where(alive, a, torch.full_like(a, -1))
            last_info = torch.where(alive, info, last_info)
            alive = alive & ~done.bool()
            if not bool(alive.any()):
                break
        self.qnetwork_local.train()
        acts_h = acts.cpu().numpy(); length_h = length.cpu().numpy()
        action_data = [[int(x) for x in acts_h[:length_h[i], i]] for i in range(n)]
        reward_data = [float(x) for x in ret.cpu().numpy()]
        success_data = [bool(x) for x in (last_info == 4).cpu().numpy()]
        time_data = [float(r0["dt"] * r0["N"] * l) for l in length_h]
        energy_data = [float(x) for x in energy.cpu().numpy()]
        self._log_evaluation(greedy, action_data, reward_data, success_data, time_data, energy_data, eval_log_path)
        return dict(rewards=reward_data, successes=success_data, times=time_data, energies=energy_data, actions=action_data)

    def _log_evaluation(self, greedy, action_data, reward_data, success_data, time_data, energy_data, eval_log_path,
                        verbose=True):
        """agent.py:367-398: summary print + append + npz with the reference's keys."""
        policy = "greedy" if greedy else "adaptive"
        if verbose:
            idx = np.where(np.array(success_data) == 1)[0]
            avg_t = np.mean(np.array(time_data)[idx]) if len(idx) else float("nan")
            avg_e = np.mean(np.array(energy_data)[idx]) if len(idx) else float("nan")
            print(f"++++++++ Evaluation info ({policy} IQN) ++++++++")
            print(f"Avg cumulative reward: {np.mean(reward_data):.2f}")
            print(f"Success rate: {np.sum(success_data) / len(success_data):.2f}")
            print(f"Avg time: {avg_t:.2f}")
            print(f"Avg energy: {avg_e:.2f}")
            print(f"++++++++ Evaluation info ({policy} IQN) ++++++++\n")
        self.eval_timesteps[policy].append(int(round(self.current_timestep * getattr(self, "_report_scale", 1.0))))
        self.eval_actions[policy].append(action_data)
        self.eval_rewards[policy].append(reward_data)
        self.eval_successes[policy].append(success_data)
        self.eval_times[policy].append(time_data)
        self.eval_energies[policy].append(energy_data)
        if eval_log_path is not None:
            filename = "greedy_evaluations.npz" if greedy else "adaptive_evaluations.npz"
            np.savez(os.path.join(eval_log_path, filename),
                     timesteps=np.array(self.eval_timesteps[policy]),
                     actions=np.array(self.eval_actions[policy], dtype=object),
                     rewards=np.array(self.eval_rewards[policy]),
                     successes=np.array(self.eval_successes[policy]),
                     times=np.array(self.eval_times[policy]),
                     energies=np.array(self.eval_energies[policy]))
