"""Direct gradient all-reduce over peer-mapped buffers, fused into the optimizer launch (csrc/peer.hip).

Slide-level data parallelism reduces one flat fp32 bucket per step (833 KB at the BRACS shape).  `train.GradBucket.allreduce_mean`
does it with `torch.distributed.all_reduce` (RCCL): a ring / tree launch that is latency-bound at this size and sits between the
last weight gradient and AdamW.  `PeerReducer` is the one-shot alternative for the ranks of ONE node:

  * every rank allocates two gradient slots (step parity) and a flag array and maps its peers'.  Round 5: the allocation is
    FINE-GRAINED / UNCACHED device memory when the runtime offers it (`hipExtMallocWithFlags(hipDeviceMallocUncached)` through ctypes
    into libamdhip64, exported with `hipIpcGetMemHandle`): flags a peer spins on and slots it reads while both kernels run must not
    sit in a non-coherent cache.  Where that is unavailable the round-4 path remains (ordinary torch allocations shared through
    torch's CUDA-IPC reductions, system-scope atomics + fences in the kernels);
  * set-up runs in PHASES with an agreed outcome after each (allocate + export -> all-reduce the status -> exchange handles -> map ->
    all-reduce the status): a rank that fails anywhere takes every rank to the next fallback together, no collective is ever entered
    by only some ranks;
  * `FlatAdamW.step` then issues `acmil_peer_publish` (bucket -> own slot with write-through stores, step flag into every peer's
    flag array) and `acmil_adamw_step_peer` (wait for the peers' flags, add the W buckets in rank order straight from the mapped
    pointers -- xGMI reads --, divide by W, AdamW): two launches, no collective, bit-identical sums on every rank;
  * the FIRST step is checked against RCCL (`FlatAdamW._first_peer_step`): the directly reduced bucket must equal one
    `all_reduce` of the same bucket; on a mismatch (stale remote lines on a fabric this path has never run on) the step is redone
    on the collective's result and every rank falls back to torch.distributed for good, with a warning.

The reference has no multi-GPU path (Step3_WSI_classification_ACMIL.py is single-process); RCCL stays the default here, the direct
path is switched on with `--dp-reduce direct` / ACMIL_DP_REDUCE=direct and falls back to RCCL when the mapping cannot be set up.
"""
from __future__ import annotations

import ctypes
import os
from typing import List, Optional

import torch

from . import _lib

PEER_MAX = 8
_HIP_UNCACHED = 0x3                 # hipDeviceMallocUncached
_HIP_IPC_LAZY_PEER = 0x1            # hipIpcMemLazyEnablePeerAccess


class _IpcHandle(ctypes.Structure):
    _fields_ = [("reserved", ctypes.c_char * 64)]


class _Hip:
    """The five runtime calls the uncached path needs, through ctypes (torch exposes none of them)."""
    _inst = None

    def __init__(self):
        self.lib = ctypes.CDLL("libamdhip64.so")
        L = self.lib
        L.hipExtMallocWithFlags.argtypes = [ctypes.POINTER(ctypes.c_void_p), ctypes.c_size_t, ctypes.c_uint]
        L.hipIpcGetMemHandle.argtypes = [ctypes.POINTER(_IpcHandle), ctypes.c_void_p]
        L.hipIpcOpenMemHandle.argtypes = [ctypes.POINTER(ctypes.c_void_p), _IpcHandle, ctypes.c_uint]
        L.hipIpcCloseMemHandle.argtypes = [ctypes.c_void_p]
        L.hipFree.argtypes = [ctypes.c_void_p]
        L.hipMemset.argtypes = [ctypes.c_void_p, ctypes.c_int, ctypes.c_size_t]
        L.hipGetLastError.argtypes = []
        for f in ("hipExtMallocWithFlags", "hipIpcGetMemHandle", "hipIpcOpenMemHandle", "hipIpcCloseMemHandle", "hipFree", "hipMemset", "hipGetLastError"):
            getattr(L, f).restype = ctypes.c_int

    @classmethod
    def get(cls) -> "_Hip":
        if cls._inst is None:
            cls._inst = cls()
        return cls._inst

    @classmethod
    def ok(cls, rc: int, what: str):
        if rc != 0:
            try:
                cls.get().lib.hipGetLastError()        # clear the runtime's sticky last-error: torch's next call would trip over it
            except Exception:
                pass
            raise RuntimeError("%s failed (hipError %d)" % (what, rc))


class PeerReducer:
    """Built by `try_create` only (its phases are collective)."""

    def __init__(self, n_total: int, device: torch.device, rank: int, world: int, group=None, timeout_s: Optional[float] = None):
        if not (1 < world <= PEER_MAX):
            raise RuntimeError("acmil_amd.PeerReducer: 2..%d ranks of one node" % PEER_MAX)
        # how long an optimizer launch waits for its peers' gradients before it gives up (RCCL would wait for ever): long enough for a
        # peer that reads a large slide from a cold disk or initialises lazily on its first step; ACMIL_PEER_TIMEOUT_S overrides
        if timeout_s is None:
            timeout_s = float(os.environ.get("ACMIL_PEER_TIMEOUT_S", "120"))
        self.n_total, self.device, self.rank, self.world, self.timeout_s = n_total, torch.device(device), rank, world, timeout_s
        self.group = group
        self.owner = None          # the FlatAdamW that runs the reduction inside its launch (set by its constructor)
        self.memory = None         # "uncached" | "torch"
        self.verified = False      # the first step has been compared with RCCL (FlatAdamW._first_peer_step)
        self.verdict = "unchecked"
        self.arrive = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.err = torch.zeros(1, dtype=torch.int32, device=self.device)
        self.step_id = 0
        self._own_ptrs: List[int] = []        # uncached: (slots, flags) device pointers this rank allocated
        self._opened: List[int] = []          # uncached: pointers opened from peers' handles
        self._keep = []                       # torch path: own and mapped tensors
        self._slot_base: List[int] = []       # per rank: address of its [2, n_total] fp32 slots
        self._flag_base: List[int] = []       # per rank: address of its [PEER_MAX] int32 flag array
        self._closed = False

    # ---- phase 1: allocate and export (local; may raise)
    def _alloc(self, memory: str):
        self.memory = memory
        nbytes_s, nbytes_f = 2 * self.n_total * 4, PEER_MAX * 4
        torch.cuda.synchronize(self.device)
        if memory == "uncached":
            hip = _Hip.get()
            with torch.cuda.device(self.device):
                ptrs, handles = [], []
                for nb in (nbytes_s, nbytes_f):
                    p = ctypes.c_void_p()
                    _Hip.ok(hip.lib.hipExtMallocWithFlags(ctypes.byref(p), max(nb, 4096), _HIP_UNCACHED), "hipExtMallocWithFlags(uncached)")
                    self._own_ptrs.append(p.value)
                    _Hip.ok(hip.lib.hipMemset(p, 0, max(nb, 4096)), "hipMemset")
                    h = _IpcHandle()
                    _Hip.ok(hip.lib.hipIpcGetMemHandle(ctypes.byref(h), p), "hipIpcGetMemHandle")
                    ptrs.append(p.value); handles.append(ctypes.string_at(ctypes.byref(h), 64))      # (h.reserved would stop at the first NUL)
            torch.cuda.synchronize(self.device)
            self._mine = (ptrs[0], ptrs[1])
            return ("uncached", handles[0], handles[1])
        from torch.multiprocessing.reductions import reduce_tensor
        slots = torch.zeros(2, self.n_total, dtype=torch.float32, device=self.device)
        flags = torch.zeros(PEER_MAX, dtype=torch.int32, device=self.device)
        torch.cuda.synchronize(self.device)
        self._keep += [slots, flags]
        self._mine = (slots.data_ptr(), flags.data_ptr())
        return ("torch", reduce_tensor(slots), reduce_tensor(flags))

    # ---- phase 2: map the peers' allocations (local; may raise)
    def _map(self, gathered):
        self._slot_base, self._flag_base = [], []
        for r, (kind, hs, hf) in enumerate(gathered):
            if r == self.rank:
                self._slot_base.append(self._mine[0]); self._flag_base.append(self._mine[1])
                continue
            if kind == "uncached":
                hip = _Hip.get()
                with torch.cuda.device(self.device):
                    out = []
                    for raw in (hs, hf):
                        h = _IpcHandle()
                        ctypes.memmove(ctypes.byref(h), raw, 64)
                        p = ctypes.c_void_p()
                        _Hip.ok(hip.lib.hipIpcOpenMemHandle(ctypes.byref(p), h, _HIP_IPC_LAZY_PEER), "hipIpcOpenMemHandle")
                        self._opened.append(p.value)
                        out.append(p.value)
                self._slot_base.append(out[0]); self._flag_base.append(out[1])
            else:
                fs, args_s = hs
                ff, args_f = hf
                ts, tf = fs(*args_s), ff(*args_f)          # mapped into this process
                self._keep += [ts, tf]
                self._slot_base.append(ts.data_ptr()); self._flag_base.append(tf.data_ptr())
        w = self.world
        self._flag_ptrs = (ctypes.c_void_p * w)(*self._flag_base)
        self._slot_ptrs = [(ctypes.c_void_p * w)(*[b + par * self.n_total * 4 for b in self._slot_base]) for par in (0, 1)]
        self.my_flags_ptr = self._flag_base[self.rank]

    def _release(self):
        """Local release of whatever phases 1 / 2 acquired (no collective)."""
        if self._closed:
            return
        self._closed = True
        try:
            hip = _Hip.get() if (self._opened or self._own_ptrs) else None
            for p in self._opened:
                hip.lib.hipIpcCloseMemHandle(ctypes.c_void_p(p))
            for p in self._own_ptrs:
                hip.lib.hipFree(ctypes.c_void_p(p))
            if hip is not None:
                hip.lib.hipGetLastError()
        except Exception:
            pass
        self._opened, self._own_ptrs, self._keep = [], [], []

    @staticmethod
    def _agree(ok: int, device, group) -> bool:
        import torch.distributed as dist
        flag = torch.tensor([ok], dtype=torch.int32, device=device if dist.get_backend(group) == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN, group=group)
        return int(flag.item()) == 1

    @classmethod
    def try_create(cls, n_total, device, rank, world, group=None, memories=("uncached", "torch")) -> Optional["PeerReducer"]:
        """The reducer, or None when the peers cannot be mapped -- the same answer on every rank: each phase ends in an all-reduce of its
        outcome, and every rank enters every collective (a rank whose phase failed contributes its failure, it never skips ahead)."""
        import torch.distributed as dist
        for memory in memories:
            red, ok, export = None, 1, None
            try:
                red = cls(n_total, device, rank, world, group)
                export = red._alloc(memory)
            except Exception as e:      # e.g. a runtime without uncached allocations / IPC export
                ok = 0
                print("acmil_amd.PeerReducer[%d]: %s slots unavailable (%s: %s)" % (rank, memory, type(e).__name__, e))
            if not cls._agree(ok, device, group):
                if red is not None:
                    red._release()
                continue
            gathered = [None] * world
            dist.all_gather_object(gathered, export, group=group)          # everybody is here: phase 1 was agreed
            try:
                red._map(gathered)
            except Exception as e:      # e.g. a driver without IPC support, ranks on different nodes
                ok = 0
                print("acmil_amd.PeerReducer[%d]: mapping the peers' %s slots failed (%s: %s)" % (rank, memory, type(e).__name__, e))
            if cls._agree(ok, device, group):
                return red
            # somebody could not map: nobody may free before everybody has stopped trying
            dist.barrier(group=group)
            red._release()
        return None

    def close(self):
        """Collective teardown: a peer's last optimizer launch may still be READING this rank's slots -- drain the stream, meet the
        peers, then unmap / free.  Safe to call twice."""
        if self._closed:
            return
        import torch.distributed as dist
        torch.cuda.synchronize(self.device)
        try:
            if dist.is_initialized():
                dist.barrier(group=self.group)
        except Exception:
            pass
        self._release()

    def __del__(self):
        try:
            self._release()
        except Exception:
            pass

    def publish(self, bucket: torch.Tensor) -> int:
        """Enqueue: bucket [n_total] -> this rank's slot of the next step, flags raised at every peer.  Returns the step id."""
        if bucket.numel() != self.n_total or bucket.dtype != torch.float32 or not bucket.is_contiguous() or bucket.device != self.device:
            raise RuntimeError("acmil_amd.PeerReducer.publish: the flat fp32 bucket of this reducer, on its device")
        self.step_id += 1
        my_slot = self._slot_base[self.rank] + (self.step_id & 1) * self.n_total * 4
        rc = _lib.load().acmil_peer_publish(bucket.data_ptr(), my_slot, self.n_total, self._flag_ptrs,
                                            self.world, self.rank, self.step_id, self.arrive.data_ptr(),
                                            torch.cuda.current_stream(self.device).cuda_stream)
        _lib.check(rc, "acmil_peer_publish")
        return self.step_id

    def slot_ptrs(self):
        return self._slot_ptrs[self.step_id & 1]

    def check(self):
        """Host-side look at the timeout word (synchronises): raises if a peer's flag never arrived.  A reducer the optimizer has
        given up (first-step check failed: training continues on torch.distributed) has nothing left to report."""
        if self.owner is None and self.verdict.startswith("mismatch"):
            return
        if int(self.err.item()) != 0:
            raise RuntimeError("acmil_amd.PeerReducer: a peer's gradient flag did not arrive within %.1f s (rank %d)" % (self.timeout_s, self.rank))


def dp_reduce_mode(default: str = "rccl") -> str:
    v = os.environ.get("ACMIL_DP_REDUCE", default).lower()
    if v not in ("rccl", "direct"):
        raise RuntimeError("ACMIL_DP_REDUCE must be 'rccl' or 'direct'")
    return v
