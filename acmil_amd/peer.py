"""Direct gradient all-reduce over peer-mapped buffers, fused into the optimizer launch (csrc/peer.hip).

Slide-level data parallelism reduces one flat fp32 bucket per step (833 KB at the BRACS shape).  `train.GradBucket.allreduce_mean`
does it with `torch.distributed.all_reduce` (RCCL): a ring / tree launch that is latency-bound at this size and sits between the
last weight gradient and AdamW.  `PeerReducer` is the one-shot alternative for the ranks of ONE node:

  * every rank allocates two gradient slots (step parity) and a flag array, and maps its peers' with torch's CUDA-IPC tensor
    sharing (`torch.multiprocessing.reductions`; the handles travel through `torch.distributed.all_gather_object` on whatever
    backend the job runs -- gloo in the single-GPU test, nccl in a real job);
  * `FlatAdamW.step` then issues `acmil_peer_publish` (bucket -> own slot with write-through stores, step flag into every peer's
    flag array) and `acmil_adamw_step_peer` (wait for the peers' flags, add the W buckets in rank order straight from the mapped
    pointers -- xGMI reads --, divide by W, AdamW): two launches, no collective, bit-identical sums on every rank.

The reference has no multi-GPU path (Step3_WSI_classification_ACMIL.py is single-process); RCCL stays the default here, the direct
path is switched on with `--dp-reduce direct` / ACMIL_DP_REDUCE=direct and falls back to RCCL when the mapping cannot be set up.
"""
from __future__ import annotations

import ctypes
import os
from typing import Optional

import torch

from . import _lib

PEER_MAX = 8


class PeerReducer:
    def __init__(self, n_total: int, device: torch.device, rank: int, world: int, group=None, timeout_s: Optional[float] = None):
        """n_total = elements of the flat bucket (gradients + the range-flag slot).  Collective: every rank of `group` must call it."""
        import torch.distributed as dist
        from torch.multiprocessing.reductions import reduce_tensor
        if not (1 < world <= PEER_MAX):
            raise RuntimeError("acmil_amd.PeerReducer: 2..%d ranks of one node" % PEER_MAX)
        # how long an optimizer launch waits for its peers' gradients before it gives up (RCCL would wait for ever): long enough for a
        # peer that reads a large slide from a cold disk or initialises lazily on its first step; ACMIL_PEER_TIMEOUT_S overrides
        if timeout_s is None:
            timeout_s = float(os.environ.get("ACMIL_PEER_TIMEOUT_S", "120"))
        self.n_total, self.device, self.rank, self.world, self.timeout_s = n_total, torch.device(device), rank, world, timeout_s
        self.owner = None          # the FlatAdamW that runs the reduction inside its launch (set by its constructor)
        # [slot 0 | slot 1] fp32 and the flag array live in their own allocations (an IPC handle covers a whole allocation)
        self.slots = torch.zeros(2, n_total, dtype=torch.float32, device=self.device)
        self.flags = torch.zeros(PEER_MAX, dtype=torch.int32, device=self.device)
        self.arrive = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        torch.cuda.synchronize(self.device)
        mine = (reduce_tensor(self.slots), reduce_tensor(self.flags))
        gathered = [None] * world
        dist.all_gather_object(gathered, mine, group=group)
        self._peer_slots, self._peer_flags = [], []
        for r, (hs, hf) in enumerate(gathered):
            if r == rank:
                self._peer_slots.append(self.slots); self._peer_flags.append(self.flags)
            else:
                fs, args_s = hs
                ff, args_f = hf
                self._peer_slots.append(fs(*args_s)); self._peer_flags.append(ff(*args_f))      # mapped into this process
        # (no barrier here: a rank whose mapping failed has left through an exception, and a collective it never enters would hang the
        #  others -- try_create's all-reduce of the outcome is the rendezvous; the slots stay alive as long as this object does)
        self._flag_ptrs = (ctypes.c_void_p * world)(*[t.data_ptr() for t in self._peer_flags])
        self._slot_ptrs = [(ctypes.c_void_p * world)(*[t[par].data_ptr() for t in self._peer_slots]) for par in (0, 1)]
        self.step_id = 0

    @classmethod
    def try_create(cls, n_total, device, rank, world, group=None) -> Optional["PeerReducer"]:
        """The reducer, or None when the peers cannot be mapped (then every rank gets None: the decision is all-reduced)."""
        import torch.distributed as dist
        ok, red = 1, None
        try:
            red = cls(n_total, device, rank, world, group)
        except Exception as e:      # e.g. a driver without IPC support, ranks on different nodes
            ok = 0
            print("acmil_amd.PeerReducer: direct all-reduce unavailable on rank %d (%s: %s); using torch.distributed" % (rank, type(e).__name__, e))
        flag = torch.tensor([ok], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return red if int(flag.item()) == 1 else None

    def publish(self, bucket: torch.Tensor) -> int:
        """Enqueue: bucket [n_total] -> this rank's slot of the next step, flags raised at every peer.  Returns the step id."""
        if bucket.numel() != self.n_total or bucket.dtype != torch.float32 or not bucket.is_contiguous() or bucket.device != self.device:
            raise RuntimeError("acmil_amd.PeerReducer.publish: the flat fp32 bucket of this reducer, on its device")
        self.step_id += 1
        rc = _lib.load().acmil_peer_publish(bucket.data_ptr(), self.slots[self.step_id & 1].data_ptr(), self.n_total, self._flag_ptrs,
                                            self.world, self.rank, self.step_id, self.arrive.data_ptr(),
                                            torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "acmil_peer_publish")
        return self.step_id

    def slot_ptrs(self):
        return self._slot_ptrs[self.step_id & 1]

    def check(self):
        """Host-side look at the timeout word (synchronises): raises if a peer's flag never arrived."""
        if int(self.err.item()) != 0:
            raise RuntimeError("acmil_amd.PeerReducer: a peer's gradient flag did not arrive within %.1f s (rank %d)" % (self.timeout_s, self.rank))


def dp_reduce_mode(default: str = "rccl") -> str:
    v = os.environ.get("ACMIL_DP_REDUCE", default).lower()
    if v not in ("rccl", "direct"):
        raise RuntimeError("ACMIL_DP_REDUCE must be 'rccl' or 'direct'")
    return v
