"""One-shot gradient exchange of a shared learner over IPC-mapped mailboxes (C-ABI `mn_xchg_*`, csrc/iqn_train.hip: iqn_grad_gather).

`IQNAgent(distributed=True)` all-reduces its flat 143 KB gradient over RCCL by default (`iqn/fused_train.py`).  With
`agent.exchange = "mailbox"` the reduction + Adam blocks of every rank's gradient step publish their reduced columns into a mailbox in the rank's own
HBM (uncached device memory) and gather the same columns of every rank, in rank order, reading the peers' over IPC-mapped pointers (xGMI between
GPUs; two ranks sharing one GPU is how the tests run it) -- inside the step's own launch, no collective.  The process group is used once, to exchange the
64-byte IPC handles.  A gather that does not get a peer's granules within the bound skips the update and counts itself: `FusedTrainer.check_timeouts`
turns that into an exception at the loop's evaluation points.
"""
import ctypes as C
import os
import weakref

import torch

from .. import _capi


class MailboxExchange:
    def __init__(self, device, rank=None, world=None, group=None):
        import torch.distributed as dist
        self.device = torch.device(device)
        self.rank = dist.get_rank(group) if rank is None else int(rank)
        self.world = dist.get_world_size(group) if world is None else int(world)
        L = _capi.lib()
        h = C.c_void_p()
        with torch.cuda.device(self.device):
            rc = L.mn_xchg_create(self.rank, self.world, C.byref(h))
            if rc:
                raise _capi.MarineNavHipError(f"mn_xchg_create failed ({rc})")
            self.h = h
            self._fin = weakref.finalize(self, L.mn_xchg_destroy, h)
            mine = C.create_string_buffer(64)
            if self.world > 1:
                rc = L.mn_xchg_export(h, mine)
                if rc:
                    raise _capi.MarineNavHipError(f"mn_xchg_export failed ({rc}): {L.mn_xchg_last_error(h).decode()}")
                handles = [None] * self.world
                dist.all_gather_object(handles, bytes(mine.raw), group=group)
                for r, hb in enumerate(handles):
                    if r != self.rank:
                        rc = L.mn_xchg_import(h, r, C.create_string_buffer(hb, 64))
                        if rc:
                            raise _capi.MarineNavHipError(f"mn_xchg_import of rank {r}'s mailbox failed ({rc}): {L.mn_xchg_last_error(h).decode()}")
        self._attached = set()
        ms = os.environ.get("MN_XCHG_TIMEOUT_MS")      # how long a gather waits for a peer (default: 30 s with peers, 2 s alone)
        if ms:
            self.set_timeout_ms(int(ms))

    def set_timeout_ms(self, ms):
        rc = _capi.lib().mn_xchg_set_timeout_ms(self.h, int(ms))
        if rc:
            raise _capi.MarineNavHipError(f"mn_xchg_set_timeout_ms failed ({rc})")

    def memory_kind(self):
        """How the mailbox was allocated: "uncached" / "fine-grained" device memory (visible to peers inside a running kernel), or "coarse" (plain hipMalloc)."""
        return {2: "uncached", 1: "fine-grained", 0: "coarse"}.get(_capi.lib().mn_xchg_memory_kind(self.h), "?")

    def attach(self, workspace, batch):
        """The learner stepping on `workspace` publishes its reduced gradient into this rank's mailbox from now on."""
        key = (workspace.data_ptr(), int(batch))
        if key in self._attached:
            return
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = _capi.lib().mn_xchg_attach(self.h, C.c_void_p(workspace.data_ptr()), int(batch), stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_xchg_attach failed ({rc})")
        self._attached.add(key)

    def detach(self, workspace, batch):
        key = (workspace.data_ptr(), int(batch))
        if key in self._attached:
            stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
            rc = _capi.lib().mn_xchg_attach(None, C.c_void_p(workspace.data_ptr()), int(batch), stream)
            if rc:
                raise _capi.MarineNavHipError(f"mn_xchg_attach(NULL) failed ({rc})")
            self._attached.discard(key)

    def exchange(self, grad, workspace, batch, grad_scale):
        """grad := sum over ranks (rank order), norm partials of grad_scale * grad into the workspace; on the current stream."""
        stream = C.c_void_p(torch.cuda.current_stream(self.device).cuda_stream)
        rc = _capi.lib().mn_iqn_train_exchange(self.h, C.c_void_p(grad.data_ptr()), C.c_void_p(workspace.data_ptr()), int(batch),
                                               C.c_float(grad_scale), stream)
        if rc:
            raise _capi.MarineNavHipError(f"mn_iqn_train_exchange failed ({rc})")

    def timeouts(self):
        n = C.c_int32()
        torch.cuda.synchronize(self.device)
        rc = _capi.lib().mn_xchg_status(self.h, C.byref(n))
        if rc:
            raise _capi.MarineNavHipError(f"mn_xchg_status failed ({rc})")
        return n.value
