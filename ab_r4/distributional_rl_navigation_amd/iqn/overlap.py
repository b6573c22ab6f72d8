"""The vectorised loop of `IQNAgent.vec_step` over H sub-batches of the environments on H HIP streams, with the learner in
the shadow of the actors.

Why: a vector step is  act (one compute-bound kernel, ~85 % of the time)  ->  env step + replay append  ->  reset of finished
envs  (-> gradient steps), and everything but the act kernel is a chain of short, latency-bound launches that leave the chip
idle behind a kernel that fills it.  With the envs in H = 2 handles (shard h = global env indices [h n / H, (h + 1) n / H):
the same worlds and episodes as one handle of n) each sub-batch runs its own act -> step_append -> reset_done chain
on its own stream, free-running: while one half's env kernels -- and, on the last stream, the gradient steps -- wait on memory,
the other half's act workgroups have the CUs.  The act kernel is launched with a finer grid for this (`mn_iqn_set_grid`: CUs
are handed back every few tens of microseconds instead of after the whole launch).

What stays ordered (events; none of them normally stalls a stream):
  * a sub-batch's own chain (its stream);
  * replay appends: the ring pointer advances on the host in issue order, so the H appends of a vector step write disjoint
    slots; an append also waits for the other sub-batches' PREVIOUS append, so a slot is never rewritten out of order;
  * training events run on the LAST sub-batch's stream, after every sub-batch's append of that vector step; the other
    sub-batches' NEXT append waits for them (a gradient step never gathers from rows that are being overwritten);
  * weights: the learner writes them while the other streams' act kernels are in flight -- those read a packed weight IMAGE, and
    there are two (`mn_iqn_pack_slot`): after its gradient steps the learner packs the image it is not using and switches; the
    other sub-batches finish the vector step they are in on the previous image and switch with the next one, i.e. **an actor
    lags the learner by at most one vector step** (the reference's single loop has no such lag; with 65 536 envs per GPU a
    vector step is 0.02 % of a run's experience).  Exact-f32 act variants have one image: there the streams are joined around a
    training event instead;
  * tau / exploration draws: one counter-based generator PER sub-batch (two concurrent act calls must not share a draw buffer).
Same reference semantics as `vec_step` (agent.py:113-171 per env) otherwise; the replay ring holds the same transitions in a
different row order.
"""
import torch

from .cadence import cadence_tick
from .fused_act import ActRng, act_context, fused_act


class SplitBatchLoop:
    def __init__(self, agent, envs, act_grid=1024, order_appends=True, learner_stream=False):
        assert len(envs) >= 1 and all(e.device == envs[0].device for e in envs)
        self.agent, self.envs = agent, list(envs)
        self.device = envs[0].device
        assert self.device.type == "cuda" and agent.use_fused_act, "SplitBatchLoop drives the fused HIP act kernel"
        assert not getattr(agent, "shared_taus", False), "launch-shared taus read the live layer-1 weights: not with the double-buffered weight image"
        # step_append writes 1-step transitions; the learner would discount them with GAMMA ** n_step
        assert agent.n_step == 1, "SplitBatchLoop appends 1-step transitions (mn_step_append): n_step must be 1"
        self.n_envs = sum(e.n_envs for e in envs)
        H = len(envs)
        self.streams = [torch.cuda.Stream(device=self.device) for _ in range(H)]
        self.rngs = [ActRng(agent.gen.initial_seed() + 7919 * (h + 1), self.device) for h in range(H)]
        self.append_done = [torch.cuda.Event() for _ in range(H)]
        self.fork, self.learn_done = torch.cuda.Event(), torch.cuda.Event()
        self.obs = [None] * H
        self.ctx = act_context(agent.qnetwork_local)
        if H > 1:
            self.ctx.set_grid(act_grid)
        self.slots = self.ctx.variant == 2          # two explicitly managed weight images (split-f16 kernel only)
        self.cur_slot, self.slot_of = 0, [0] * H
        self.switch_pending = [False] * H
        self._appended_once = False
        self.order_appends = order_appends
        # learner_stream: the gradient steps get a stream of their own instead of following the last sub-batch's env kernels, so
        # that EVERY sub-batch's next act runs beside them (pays when a training event is longer than an act launch)
        self.lstream = torch.cuda.Stream(device=self.device) if (learner_stream and self.slots) else None

    def reset(self):
        main = torch.cuda.current_stream(self.device)
        self.obs = [e.reset() for e in self.envs]
        if self.slots:
            self.ctx.pack_slot(self.agent.qnetwork_local, 0)
            self.cur_slot, self.slot_of = 0, [0] * len(self.envs)
        else:
            self.ctx.refresh(self.agent.qnetwork_local)
        self.fork.record(main)
        for s in self.streams:
            s.wait_event(self.fork)
        return self.obs

    def close(self):
        """Back to the agent's ordinary act path (cached image, persistent grid)."""
        self.join()
        if self.slots:
            self.ctx.select_slot(-1)
        self.ctx.invalidate()
        self.ctx.set_grid(0)

    def step(self, eps, cvar=1.0, train_every=None, per_iter=None):
        """One vector step of all sub-batches (+ the training event the cadence asks for).  Returns (obs list, reward list,
        done list, info list, loss or None); the per-sub-batch tensors are owned by the env handles and valid on their streams
        (call `join()` before reading them from the calling stream)."""
        ag = self.agent
        train_every = ag.UPDATE_EVERY if train_every is None else train_every
        per_iter = self.n_envs if per_iter is None else per_iter
        H = len(self.envs)
        outs = []
        for h, (e, s) in enumerate(zip(self.envs, self.streams)):
            with torch.cuda.stream(s):
                if self.slots:
                    self.ctx.select_slot(self.slot_of[h])
                a = fused_act(ag.qnetwork_local, self.obs[h], eps, cvar, rng=self.rngs[h])
                if self.switch_pending[h]:      # a training event ran on another stream while this one was acting
                    s.wait_event(self.learn_done)
                    self.slot_of[h] = self.cur_slot
                    self.switch_pending[h] = False
                if self._appended_once and self.order_appends:
                    for k in range(H):
                        if k != h:
                            s.wait_event(self.append_done[k])      # the other sub-batches' PREVIOUS appends (normally long done)
                nxt, reward, done, info = e.step_append(a, self.obs[h], ag.memory)
                self.append_done[h].record(s)
                self.obs[h] = e.reset_done()
                outs.append((reward, done, info))
        self._appended_once = True
        loss = None
        due = cadence_tick(ag, train_every)      # iqn/cadence.py: the one statement of the loop's cadence
        if due.train or due.sync:
            loss = self._training_event(due.train, due.sync)
        if ag.current_timestep >= ag.learning_starts:
            ag.learning_timestep += 1
        ag.current_timestep += per_iter
        return self.obs, [o[0] for o in outs], [o[1] for o in outs], [o[2] for o in outs], loss

    def _training_event(self, train_now, sync_now):
        ag, H = self.agent, len(self.envs)
        loss = None
        if self.slots:
            L = H - 1 if self.lstream is None else -1
            s = self.streams[L] if self.lstream is None else self.lstream
            with torch.cuda.stream(s):
                for k in range(H):
                    if k != L:
                        s.wait_event(self.append_done[k])
                if train_now:
                    loss = ag.train_steps_from_memory(ag.grad_steps_per_update)
                if sync_now:
                    ag._sync_target()
                new = self.cur_slot ^ 1
                if train_now:
                    self.ctx.pack_slot(ag.qnetwork_local, new)
                self.learn_done.record(s)
            if train_now:
                self.cur_slot = new
                if L >= 0:
                    self.slot_of[L] = new
            for k in range(H):
                if k != L:
                    self.switch_pending[k] = True
            return loss
        main = torch.cuda.current_stream(self.device)      # one image: join, train on the calling stream, refresh, fork
        for ev in self.append_done:
            main.wait_event(ev)
        if train_now:
            loss = ag.train_steps_from_memory(ag.grad_steps_per_update)
        if sync_now:
            ag._sync_target()
        self.ctx.refresh(ag.qnetwork_local)
        self.fork.record(main)
        for s in self.streams:
            s.wait_event(self.fork)
        return loss

    def join(self):
        """Make the calling stream wait for everything the sub-batch streams were given."""
        main = torch.cuda.current_stream(self.device)
        for s in self.streams + ([self.lstream] if self.lstream is not None else []):
            main.wait_stream(s)
