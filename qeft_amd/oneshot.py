"""One-shot all-reduce for the tensor-parallel decode path (kernel: csrc/oneshot.hip; no reference counterpart).

A decode token makes two all-reduces per layer of an fp32 [hidden] vector (16 KB at hidden 4096) -- latency-bound on xGMI, where a
ring collective pays its 2 (P - 1) hops and its launch machinery for a payload a link moves in ~0.1 us.  `OneShotAllReduce` owns
one MAILBOX per rank (device memory, exported with hipIpcGetMemHandle and opened by every peer), and `all_reduce(t)` is ONE kernel
launch per rank: direct peer writes of tagged granules, a poll of the local mailbox, a sum in rank order (bit-identical on every
rank).  The torch.distributed group is used ONCE, at construction, to exchange the 64-byte IPC handles; no collective of the
group runs afterwards, and the call is hipGraph-capturable.
"""
import ctypes

import torch

from . import _lib


class OneShotAllReduce:
    def __init__(self, n, device, group=None):
        import torch.distributed as dist
        self.lib = _lib.lib()
        self.group = group
        self.world = dist.get_world_size(group)
        self.rank = dist.get_rank(group)
        if self.world > self.lib.qeft_oneshot_max_world():
            raise ValueError(f"one-shot all-reduce is built for up to {self.lib.qeft_oneshot_max_world()} ranks, got {self.world}")
        self.n = int(n)
        self.device = torch.device(device)
        # Every rank runs the SAME two collectives of the group whatever happens locally (a rank that raised before a collective its
        # peers are waiting in would hang them): failures are collected, exchanged, and raised on every rank alike.
        self.box, self._opened, err = None, [], None
        mine = None
        with torch.cuda.device(self.device):
            try:
                # the mailbox is a hipMalloc block of its own (an IPC handle exports the whole allocation a pointer lives in: a
                # tensor carved out of torch's caching allocator would arrive in the peer at an offset nobody knows)
                box = ctypes.c_void_p()
                _lib.check(self.lib.qeft_oneshot_mailbox_alloc(self.world, self.n, ctypes.byref(box)))
                self.box = box.value
                self.seq = torch.zeros(1, dtype=torch.int32, device=self.device)
                self.status = torch.zeros(2, dtype=torch.int32, device=self.device)
                torch.cuda.synchronize(self.device)
                handle = ctypes.create_string_buffer(64)
                _lib.check(self.lib.qeft_oneshot_ipc_export(self.box, handle))
                mine = bytes(handle.raw)
            except Exception as e:          # noqa: BLE001 -- reported to every rank below
                err = f"rank {self.rank}: {type(e).__name__}: {e}"
            everyone = [None] * self.world
            dist.all_gather_object(everyone, mine, group=group)
            ptrs = (ctypes.c_void_p * self.world)()
            if err is None and any(h is None for h in everyone):
                err = f"rank {self.rank}: a peer has no mailbox to export"
            if err is None:
                try:
                    for r, h in enumerate(everyone):
                        if r == self.rank:
                            ptrs[r] = self.box
                            continue
                        p = ctypes.c_void_p()
                        _lib.check(self.lib.qeft_oneshot_ipc_open(ctypes.create_string_buffer(h, 64), ctypes.byref(p)))
                        self._opened.append(p)
                        ptrs[r] = p.value
                except Exception as e:      # noqa: BLE001
                    err = f"rank {self.rank}: {type(e).__name__}: {e}"
            self.ptrs = ptrs
        # doubles as the barrier: nobody writes into a mailbox that is not mapped everywhere yet
        errs = [None] * self.world
        dist.all_gather_object(errs, err, group=group)
        errs = [e for e in errs if e]
        if errs:
            self.close()
            raise RuntimeError("one-shot all-reduce could not be set up: " + "; ".join(errs)[:300])

    def all_reduce(self, t):
        """In-place fp32 sum of `t` (n elements) over the group: one kernel on the current stream."""
        assert t.dtype == torch.float32 and t.is_contiguous() and t.numel() == self.n and t.device == self.device
        _lib.check(self.lib.qeft_oneshot_allreduce_f32(t.data_ptr(), self.n, self.ptrs, self.rank, self.world, self.seq.data_ptr(),
                                                       self.status.data_ptr(), torch.cuda.current_stream(self.device).cuda_stream))

    def check_status(self):
        """Raises if any wait of an earlier call gave up (a peer never wrote: dead, or out of step)."""
        st = int(self.status[0].item())
        if st:
            raise RuntimeError(f"one-shot all-reduce: a wait gave up (status {st:#x}: slot {st & 0xff})")

    def close(self):
        for p in getattr(self, "_opened", []):
            self.lib.qeft_oneshot_ipc_close(p)
        self._opened = []
        if getattr(self, "box", None):
            self.lib.qeft_oneshot_mailbox_free(self.box)
            self.box = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
