"""Host side of the fused HIP gradient step (csrc/iqn_train.hip): IQNAgent.train (thirdparty/IQN/agent.py:269-304)
as three launches (forward / backward with in-launch target hand-off, reduction, Adam) behind `mn_iqn_train_grad*` /
`mn_iqn_train_adam`.

The kernels work on FLAT parameter vectors (35 785 floats, `named_parameters()` order).  `FusedTrainer` allocates one
flat buffer per network and re-points every `nn.Parameter` at a view of it, so the PyTorch modules (checkpoints,
`soft_update`, the fused act kernel, eager evaluation) and the HIP step always see the same memory.  The Adam
moments live in two more flat buffers that are ALSO `agent.optimizer`'s `exp_avg` / `exp_avg_sq` state (views), and the
step counter is copied between the device counter of the HIP step and the optimizer's per-parameter `step` whenever
the agent switches between the HIP and the PyTorch gradient step (`sync_to_optimizer` / `sync_from_optimizer`): one
optimizer state, whichever path runs.
"""
import ctypes as C
import os

import torch

from .. import _capi
from .fused_act import weights_changed

P_TOTAL = 35785
_ORDER = ("velocity_encoder", "goal_encoder", "sensor_encoder", "cos_embedding", "hidden_layer", "hidden_layer_2",
          "output_layer")


def _p(t):
    return C.c_void_p(t.data_ptr())


def flatten_network(net):
    """One contiguous float32 buffer holding all parameters of `net`; the parameters become views of it."""
    names = [n for n, _ in net.named_parameters()]
    assert names == [f"{m}.{s}" for m in _ORDER for s in ("weight", "bias")], names
    params = list(net.parameters())
    dev = params[0].device
    assert dev.type == "cuda", "the fused gradient step is a HIP kernel: parameters must live on the GPU"
    flat = torch.empty(sum(p.numel() for p in params), dtype=torch.float32, device=dev)
    assert flat.numel() == P_TOTAL
    off = 0
    with torch.no_grad():
        for p in params:
            n = p.numel()
            flat[off:off + n].copy_(p.detach().reshape(-1))
            p.data = flat[off:off + n].view(p.shape)
            off += n
    return flat


class FusedTrainer:
    def __init__(self, agent):
        self.agent = agent
        self.device = agent.device
        self.local = flatten_network(agent.qnetwork_local)
        self.target = flatten_network(agent.qnetwork_target)
        z = lambda: torch.zeros(P_TOTAL, dtype=torch.float32, device=self.device)
        self.grad, self.exp_avg, self.exp_avg_sq = z(), z(), z()
        self.step_dev = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.loss = torch.zeros(1, dtype=torch.float32, device=self.device)
        self._ws, self._ws_batch, self._ws_by_batch = None, 0, {}
        self._mailbox = None         # iqn/mailbox.py: MailboxExchange of a shared learner with agent.exchange == "mailbox"
        self._staged_key = None      # (ring, its version, rows, batch, workspace) the workspace holds a staged next batch for
        self._graph, self._graph_key = None, None
        # {seed, call counter} of the sampling kernel (same seed family as the replay memory's generator)
        self.rng_state = torch.tensor([int(agent.memory.gen.initial_seed()) & 0x7FFFFFFFFFFFFFFF, 0], dtype=torch.int64,
                                      device=self.device)
        self.rng_state[0] += 0x9E3779B1 * int(getattr(agent, "rank", 0))     # decorrelate the ranks of a shared learner
        self._idx, self._taus = {}, {}
        self._arange = {}
        off = 0
        for p in agent.qnetwork_local.parameters():      # p.grad = the (clipped) gradient of the last step, as torch
            p.grad = self.grad[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self._nets = (agent.qnetwork_local, agent.qnetwork_target)
        self._adopt_optimizer_state(agent.optimizer)

    # ---- one Adam state for both gradient-step paths ---------------------------------------------------------------
    def _adopt_optimizer_state(self, opt):
        """Make `opt.state[p]['exp_avg' / 'exp_avg_sq']` views of the flat moment buffers (keeping what the optimizer
        had accumulated so far) and take over its step count."""
        step = 0
        off = 0
        for p in self.agent.qnetwork_local.parameters():
            n = p.numel()
            st = opt.state.get(p, None)
            m_view = self.exp_avg[off:off + n].view(p.shape)
            v_view = self.exp_avg_sq[off:off + n].view(p.shape)
            if st is not None and "exp_avg" in st:
                m_view.copy_(st["exp_avg"]); v_view.copy_(st["exp_avg_sq"])
                step = int(float(st["step"]))
            else:
                st = opt.state[p]
                # same kind of `step` tensor torch.optim.Adam would create lazily (device tensor for fused / capturable)
                on_dev = any(g.get("fused") or g.get("capturable") for g in opt.param_groups)
                st["step"] = torch.zeros((), dtype=torch.float32, device=p.device if on_dev else "cpu")
            st["exp_avg"], st["exp_avg_sq"] = m_view, v_view
            off += n
        self.step_dev.fill_(step)

    def sync_to_optimizer(self, opt):
        """HIP path -> PyTorch path: hand the step count to torch.optim.Adam (the moments are shared memory)."""
        t = float(int(self.step_dev.item()))
        for p in self.agent.qnetwork_local.parameters():
            opt.state[p]["step"].fill_(t)

    def sync_from_optimizer(self, opt):
        """PyTorch path -> HIP path."""
        p0 = next(iter(self.agent.qnetwork_local.parameters()))
        self.step_dev.fill_(int(float(opt.state[p0]["step"])))

    def owns(self, agent):
        return self._nets == (agent.qnetwork_local, agent.qnetwork_target)

    def _workspace(self, batch):
        if self._ws_batch != batch:
            ws = self._ws_by_batch.get(batch)
            if ws is None:
                n = _capi.lib().mn_iqn_train_workspace_floats(batch)
                if n < 0:
                    raise ValueError("fused IQN gradient step: the batch size must be even")
                # one workspace PER batch size, never re-allocated: a captured hipGraph (graphed_steps) holds its raw pointer, and the
                # TD-target hand-off tags, their epoch word, the tickets and the staged next batch live in it between calls
                ws = self._ws_by_batch[batch] = torch.empty(n, dtype=torch.float32, device=self.device)
                rc = _capi.lib().mn_iqn_train_workspace_init(_p(ws), batch, _capi.stream_ptr(self.device))
                if rc:
                    raise _capi.MarineNavHipError(f"mn_iqn_train_workspace_init failed ({rc})")
            self._ws, self._ws_batch = ws, batch
        if self.agent.distributed and getattr(self.agent, "exchange", "collective") == "mailbox":
            if self._mailbox is None:
                from .mailbox import MailboxExchange
                self._mailbox = MailboxExchange(self.device)
            self._mailbox.attach(self._ws, batch)      # (no-op once attached) the reduction kernel publishes the gradient itself
        elif self._mailbox is not None:
            self._mailbox.detach(self._ws, batch)      # back on the collective / single-learner path: stop publishing
        return self._ws

    def sample(self, ring_size, batch):
        """ReplayBuffer.sample's index draw + the step's tau draws in ONE kernel -> (idx [B] i64, taus [2, B, 8])."""
        self._staged_key = None      # the call counter moves on: a staged batch belongs to a counter that is skipped
        if batch not in self._idx:
            self._idx[batch] = torch.empty(batch, dtype=torch.int64, device=self.device)
            self._taus[batch] = torch.empty(2, batch, self.agent.N, dtype=torch.float32, device=self.device)
        idx, taus = self._idx[batch], self._taus[batch]
        stream = _capi.stream_ptr(self.device)
        rc = _capi.lib().mn_iqn_sample(int(ring_size), batch, _p(self.rng_state), _p(idx), _p(taus), taus.numel(), stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_sample failed ({rc}): need batch <= 1024 and ring_size >= batch")
        return idx, taus

    def step_sampled(self, ring, ring_size, batch, ring_version=None):
        """`self.train(self.memory.sample())` (agent.py:131-133) in one call: the batch (ReplayBuffer.sample's uniform draw without
        replacement over the first `ring_size` rows + the step's tau draws) is drawn INSIDE the forward / backward kernel from this
        trainer's generator state -- bit-identical to `sample()` followed by `step()`, one launch less.  Returns the loss.
        `ring_version` (ReplayBuffer.version): when given, every step also stages the NEXT step's batch (MN_TRAIN_STAGE_NEXT: its
        rows, transitions and taus, gathered by the reduction kernel) and a step starts from the staged batch whenever the ring
        has not been written since (MN_TRAIN_USE_STAGED) -- same batch, same result, one memory round trip at the head of the
        launch instead of three."""
        ag = self.agent
        states, actions, rewards, next_states, dones = ring
        for t in ring:
            assert t.is_cuda and t.is_contiguous()
        assert states.dtype == torch.float32 and actions.dtype == torch.int64 and dones.dtype == torch.float32
        if batch not in self._idx:
            self._idx[batch] = torch.empty(batch, dtype=torch.int64, device=self.device)
            self._taus[batch] = torch.empty(2, batch, ag.N, dtype=torch.float32, device=self.device)
        L = _capi.lib()
        stream = _capi.stream_ptr(self.device)
        flags = 0
        if ring_version is not None:
            key = (states.data_ptr(), int(ring_version), int(ring_size), batch, self._ws.data_ptr() if self._ws is not None else 0)
            flags = 2 | (1 if key == self._staged_key else 0)
        if self._two_launches():      # forward / backward, then reduction + clip + Adam in ONE launch (mn_iqn_train_step): bit-identical
            flags |= self._one_launch_flags(batch)      # MN_TRAIN_ONE_LAUNCH: ... as a third role of the SAME launch
            rc = self._step_call((states, next_states, actions, rewards, dones), ring_size, self.rng_state, None, None, None, self._idx[batch],
                                 self._taus[batch], batch, flags, stream)
        else:
            rc = L.mn_iqn_train_grad_sampled(_p(states), _p(next_states), _p(actions), _p(rewards), _p(dones), int(ring_size),
                                             _p(self.rng_state), _p(self._idx[batch]), _p(self._taus[batch]), _p(self.local), _p(self.target),
                                             _p(self._workspace(batch)), _p(self.grad), _p(self.loss), batch, ag.N,
                                             C.c_float(ag.GAMMA ** ag.n_step), flags, stream)
        self._staged_key = (states.data_ptr(), int(ring_version), int(ring_size), batch, self._ws.data_ptr()) if ring_version is not None else None
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_train_grad_sampled / mn_iqn_train_step failed ({rc}): need batch <= 1024 and ring_size >= batch")
        if self._two_launches():
            weights_changed(ag.qnetwork_local)
            return self.loss[0]
        return self._finish_step(batch)

    def launches_per_step(self, batch=None):
        """Launches one gradient step takes on this device (C-ABI mn_iqn_train_plan): 1 / 2 for the fused forms, 3 (4 with the exchange) where they do not fit;
        an RCCL shared learner: 3 + the collective."""
        ag = self.agent
        batch = ag.BATCH_SIZE if batch is None else batch
        if not self._two_launches():
            return 4 if ag.distributed and getattr(ag, "exchange", "collective") == "mailbox" else 3
        n = _capi.lib().mn_iqn_train_plan(batch, self._one_launch_flags(batch), 1 if ag.distributed else 0)
        if n < 0:
            raise _capi.MarineNavHipError(f"mn_iqn_train_plan failed ({n})")
        return n

    def timeouts(self):
        """Bounded waits that ran out since the workspaces were made: reduction + Adam blocks (status word of every workspace) + the mailbox exchange's gathers.
        Synchronises the device.  0 in a healthy run; anything else means steps were skipped (and a shared learner's ranks no longer agree)."""
        n = 0
        for batch, ws in self._ws_by_batch.items():
            i = _capi.lib().mn_iqn_train_workspace_status_word(batch)
            n += int(ws[i:i + 1].view(torch.int32).item())
        if self._mailbox is not None:
            n += self._mailbox.timeouts()
        return n

    def check_timeouts(self):
        """Raise if any bounded wait of the gradient step ran out (a peer that died or fell > the bound behind; workgroups of a fused launch that were not
        resident together).  Called where the loop synchronises anyway: evaluation points, the end of learn_vec, bench.py's legs."""
        n = self.timeouts()
        if n:
            raise _capi.MarineNavHipError(
                f"fused IQN gradient step: {n} bounded wait(s) ran out -- the affected steps updated nothing" +
                (" and this shared learner's ranks may have diverged (a peer's gradient did not arrive within the bound; MN_XCHG_TIMEOUT_MS)"
                 if self._mailbox is not None and self._mailbox.world > 1 else ""))

    def xcd_misplaced(self, batch=None):
        """Diagnostic of the one-launch step: local workgroups that did not run on XCD (block index % 8) since the workspace was made (0 expected)."""
        batch = self.agent.BATCH_SIZE if batch is None else batch
        i = _capi.lib().mn_iqn_train_workspace_misplaced_word(batch)
        return int(self._workspace(batch)[i:i + 1].view(torch.int32).item())

    def _one_launch_flags(self, batch):
        """MN_TRAIN_ONE_LAUNCH [| MN_TRAIN_TEST_MISPLACE(k) (test hook `_test_misplace`)]: ask for the FUSED step -- reduction + clip + Adam inside the forward /
        backward launch (a shared learner's mailbox exchange rides in it too).  The library takes it where it exists (batch a multiple of 16, every workgroup a CU
        of its own) and two launches otherwise.  `IQNAgent.one_launch_step`: True / False, or unset = where it is the faster form -- batches that are multiples of
        256 (31.6 vs 34.6 us per step at 256; at 32 / 64 / 128 / 192 two launches were 1 - 8 us faster in round 4: few local workgroups each summing a large share
        of their group's rows).  MN_ONE_LAUNCH=0 / 1 overrides the unset case (A / B runs)."""
        ag = self.agent
        one = getattr(ag, "one_launch_step", None)
        if one is None:
            env = os.environ.get("MN_ONE_LAUNCH")
            one = (batch % 256 == 0) if env is None else env != "0"
        if not one:
            return 0
        return 4 | (int(getattr(ag, "_test_misplace", 0)) << 4)

    def _two_launches(self):
        """True: the whole step is ONE call into the library (mn_iqn_train_step / mn_iqn_train_step_xchg: one or two launches, more only on a device too small
        for the fused forms -- `launches_per_step`) -- a single learner, and a shared learner with the mailbox exchange (the exchange happens inside the
        reduction + Adam role); with an RCCL all-reduce between the gradient and Adam the step stays three launches + the collective.
        `agent.two_launch_step = False` selects the separate launches (A / B measurements, tests)."""
        ag = self.agent
        if not getattr(ag, "two_launch_step", True):
            return False
        return not ag.distributed or getattr(ag, "exchange", "collective") == "mailbox"

    def _step_call(self, ring5, ring_size, rng, idx, tt, tl, idx_out, taus_out, batch, flags, stream):
        """mn_iqn_train_step / mn_iqn_train_step_xchg with this trainer's buffers."""
        ag, L = self.agent, _capi.lib()
        states, next_states, actions, rewards, dones = ring5
        q = lambda t: _p(t) if t is not None else None
        common = (q(states), q(next_states), q(actions), q(rewards), q(dones), int(ring_size), q(rng), q(idx), q(tt), q(tl), q(idx_out), q(taus_out),
                  _p(self.local), _p(self.target), _p(self._workspace(batch)), _p(self.grad), _p(self.loss), _p(self.exp_avg), _p(self.exp_avg_sq),
                  _p(self.step_dev), batch, ag.N, C.c_float(ag.GAMMA ** ag.n_step), flags, C.c_double(ag.LR), C.c_double(0.9), C.c_double(0.999),
                  C.c_double(1e-8), C.c_double(0.5))
        if ag.distributed:
            mb = self._mailbox
            if mb is None:
                from .mailbox import MailboxExchange
                mb = self._mailbox = MailboxExchange(self.device)
            return L.mn_iqn_train_step_xchg(mb.h, *common, C.c_float(1.0 / mb.world), stream)
        return L.mn_iqn_train_step(*common, stream)

    def graphed_steps(self, ring, ring_size, batch, n_steps):
        """`n_steps` x step_sampled as ONE hipGraph launch (captured on first use, re-captured when the ring's tensors / size, the
        batch or n_steps change): forward / backward, reduction, the shared learner's RCCL all-reduce (RCCL kernels are
        capturable) and Adam of every step, with the generator's counter, the Adam step counter and the hand-off epoch on the
        device -- a replay continues exactly where eager calls would.  The first step of a graph draws and gathers its batch in the
        launch (the ring may have been written since the last call), steps 2.. start from the batch their predecessor staged.
        Returns the loss of the last step.  Bit-identical to the eager sequence (tests)."""
        states = ring[0]
        # everything the captured launches hold a raw pointer to is part of the key
        key = (states.data_ptr(), int(ring_size), batch, int(n_steps), bool(self.agent.distributed), self._workspace(batch).data_ptr(),
               self.local.data_ptr(), self.target.data_ptr(), self.grad.data_ptr(), self._two_launches(), self._one_launch_flags(batch))
        if self._graph_key != key:
            self._graph = None
            torch.cuda.synchronize(self.device)
            # warm-up outside the capture would advance the training state: everything the steps need is allocated here instead
            if batch not in self._idx:
                self._idx[batch] = torch.empty(batch, dtype=torch.int64, device=self.device)
                self._taus[batch] = torch.empty(2, batch, self.agent.N, dtype=torch.float32, device=self.device)
            g = torch.cuda.CUDAGraph()
            with torch.cuda.graph(g):
                self._staged_key = None                  # step 0 draws and gathers in its launch; it stages step 1's batch, and so on
                for k in range(n_steps):
                    self.step_sampled(ring, ring_size, batch, ring_version=-1)      # -1: a version no ReplayBuffer ever has
            self._graph, self._graph_key = g, key
        self._graph.replay()
        self._staged_key = None      # the ring may change before the next call; the graph's first step does not use staging anyway
        weights_changed(self.agent.qnetwork_local)
        return self.loss[0]

    def step(self, ring, idx=None, taus_target=None, taus_local=None):
        """One optimizer step.  `ring` = (states [c,26] f32, actions [c,1] i64, rewards [c,1] f32, next_states [c,26] f32,
        dones [c,1] f32), all contiguous on the device; `idx` [B] i64 selects the batch rows (None: all rows in
        order).  Returns the loss (device scalar tensor; a view of a buffer that the next step overwrites)."""
        ag = self.agent
        states, actions, rewards, next_states, dones = ring
        for t in ring:
            assert t.is_cuda and t.is_contiguous()
        assert states.dtype == torch.float32 and actions.dtype == torch.int64 and dones.dtype == torch.float32
        if idx is None:
            B = states.shape[0]
            idx = self._arange.get(B)
            if idx is None:
                idx = self._arange[B] = torch.arange(B, dtype=torch.int64, device=self.device)
        B = idx.shape[0]
        if taus_target is None or taus_local is None:
            taus = torch.rand(2, B, ag.N, device=self.device)          # model.py:149, target forward first
            tt = taus[0] if taus_target is None else taus_target
            tl = taus[1] if taus_local is None else taus_local
        else:
            tt, tl = taus_target, taus_local
        tt = tt.to(self.device, torch.float32).contiguous().view(B, ag.N)
        tl = tl.to(self.device, torch.float32).contiguous().view(B, ag.N)
        L = _capi.lib()
        stream = _capi.stream_ptr(self.device)
        if self._two_launches():
            rc = self._step_call((states, next_states, actions, rewards, dones), 0, None, idx, tt, tl, None, None, B,
                                 self._one_launch_flags(B), stream)
            if rc:
                raise _capi.MarineNavHipError(f"mn_iqn_train_step failed ({rc})")
            weights_changed(ag.qnetwork_local)
            return self.loss[0]
        rc = L.mn_iqn_train_grad(_p(states), _p(next_states), _p(actions), _p(rewards), _p(dones), _p(idx), _p(tt), _p(tl),
                                 _p(self.local), _p(self.target), _p(self._workspace(B)), _p(self.grad), _p(self.loss),
                                 B, ag.N, C.c_float(ag.GAMMA ** ag.n_step), stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_train_grad failed ({rc})")
        return self._finish_step(B)

    def _finish_step(self, B):
        """Gradient in self.grad -> (shared learner: all-reduce) -> clip + Adam -> loss."""
        ag = self.agent
        L = _capi.lib()
        stream = _capi.stream_ptr(self.device)
        scale, rewritten = 1.0, 0
        if ag.distributed and getattr(ag, "exchange", "collective") == "mailbox":
            # one-shot exchange (iqn/mailbox.py): the reduction kernel published this rank's gradient already; one gather kernel sums the
            # ranks' mailboxes in rank order and leaves the norm partials -- no collective launch, no separate norm pass
            mb = self._mailbox
            if mb is None:
                from .mailbox import MailboxExchange
                mb = self._mailbox = MailboxExchange(self.device)
            scale, rewritten = 1.0 / mb.world, 2
            mb.exchange(self.grad, self._workspace(B), B, scale)
        elif ag.distributed:
            import torch.distributed as dist
            if dist.get_backend() == "gloo":
                # debugging / single-GPU multi-process tests only: gloo reduces on the host
                g = self.grad.cpu()
                dist.all_reduce(g, op=dist.ReduceOp.SUM)
                self.grad.copy_(g)
            else:
                dist.all_reduce(self.grad, op=dist.ReduceOp.SUM)        # one 143 KB bucket over RCCL/xGMI
            # the average (1 / world) is applied inside the Adam kernel: no separate launch over the bucket
            scale, rewritten = 1.0 / dist.get_world_size(), 1
        rc = L.mn_iqn_train_adam(_p(self.local), _p(self.grad), _p(self.exp_avg), _p(self.exp_avg_sq), _p(self.step_dev),
                                 _p(self._workspace(B)), B, C.c_double(ag.LR), C.c_double(0.9), C.c_double(0.999), C.c_double(1e-8), C.c_double(0.5),
                                 C.c_float(scale), rewritten, stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_train_adam failed ({rc})")
        weights_changed(ag.qnetwork_local)      # the HIP Adam kernel wrote the weights: the act path's cached image is stale
        return self.loss[0]
