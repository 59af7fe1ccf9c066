"""FlatAdamW: torch.optim.AdamW semantics over ONE flat parameter / gradient / moment buffer, one HIP launch per step
(csrc/optim.hip).  The reference builds `torch.optim.AdamW(model.parameters(), lr, weight_decay)`
(Step3_WSI_classification_ACMIL.py:139) and calls `.step()` per slide (:219); on 26 small tensors that is ~10 foreach
launches (or a 38 us fused one) inside a training step that is launch-bound for small bags.

The parameters are re-pointed to views of one flat buffer (values preserved, `state_dict()` unchanged), gradients likewise
(shared with train.GradBucket when data parallel: the all-reduce and the optimizer then work on the same buffer).
`param_groups[0]['lr']` is honoured every step, so `adjust_learning_rate` works unchanged.
"""
from __future__ import annotations

import math
from collections import deque
from typing import Callable, Iterable, List, Optional

import torch

from . import _lib


class FlatAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, grad_buffer: Optional[torch.Tensor] = None, on_step: Optional[Callable[[], None]] = None,
                 peer=None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 for p in self.params):
            raise RuntimeError("acmil_amd.FlatAdamW: CUDA fp32 parameters only")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        # one extra slot behind the gradients = the step's range flag (written by acmil_ga_train_step, all-reduced together with
        # the gradients when data parallel, read by the AdamW launch on the device: non-zero -> the step is not applied)
        self.grad = grad_buffer if grad_buffer is not None else torch.zeros(self.numel + 1, dtype=torch.float32, device=dev)
        if self.grad.numel() not in (self.numel, self.numel + 1):
            raise RuntimeError("acmil_amd.FlatAdamW: grad_buffer size mismatch")
        self.guard_flag = self.grad[self.numel:self.numel + 1] if self.grad.numel() == self.numel + 1 else None
        self._host_flags = torch.zeros(16, dtype=torch.float32).pin_memory() if self.guard_flag is not None else None
        self._pending = deque()    # (step id, event, host slot) of launches whose flag the host has not looked at yet
        self._step_id = 0
        self._launches = 0         # ordinal handed to the kernel; the device subtracts the launches it skipped (self._skipped_dev)
        self._skipped_dev = torch.zeros(1, dtype=torch.int32, device=dev)
        self.skipped_steps = 0
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.flat[off:off + n].copy_(p.reshape(-1))
                p.data = self.flat[off:off + n].view_as(p)            # same values, now a view of the flat buffer
                if grad_buffer is None or p.grad is None or p.grad.data_ptr() != self.grad[off:off + n].data_ptr():
                    p.grad = self.grad[off:off + n].view_as(p)
                off += n
        self.exp_avg = torch.zeros_like(self.flat)
        self.exp_avg_sq = torch.zeros_like(self.flat)
        self.step_count = 0
        self.param_groups = [{"params": self.params, "lr": lr, "betas": betas, "eps": eps, "weight_decay": weight_decay}]
        self.on_step = on_step
        # peer.PeerReducer (or None): step() then publishes the bucket and runs the fused wait + reduce + AdamW launch instead of
        # reading gradients that a collective has already averaged (the caller must NOT all-reduce the bucket as well)
        self.peer = peer
        if peer is not None and (grad_buffer is None or peer.n_total != self.grad.numel()):
            raise RuntimeError("acmil_amd.FlatAdamW: the peer reducer works on the shared gradient bucket (grad_buffer), same size")
        if peer is not None:
            peer.owner = self      # tells train.GradBucket.allreduce_mean that the reduction now happens inside this optimizer's launch
        self._frozen = []          # (offset, numel) ranges that receive no gradient this run: skipped like torch skips grad=None
        self.mutations = 0         # counts every change this object makes to the parameters (owners of derived buffers compare it)
        self._in_step = None
        # or an object with launch(adamw_args, skip_flag) -> bool and committed(): the owner of a packed copy of the parameters (ACMIL_GA)
        # issues the AdamW launch itself, fused with its re-pack (acmil_ga_adamw_pack); False = not now, use the plain launch
        self.pack_hook = None

    def set_frozen(self, params: Iterable[torch.nn.Parameter]):
        """Parameters that never receive a gradient (e.g. the branch head of an n_token = 1 ACMIL model, whose loss term is not
        built: Step3_WSI_classification_ACMIL.py:201-204).  torch.optim.AdamW skips a parameter whose .grad is None -- no
        moment update and NO weight decay; the flat kernel touches every element, so these ranges are put back after the step."""
        ids = {id(p) for p in params}
        self._frozen, off = [], 0
        for p in self.params:
            if id(p) in ids:
                self._frozen.append((off, p.numel()))
            off += p.numel()

    def zero_grad(self, set_to_none: bool = False):
        self.grad.zero_()

    # ---- the update issued INSIDE the training step's own library call (acmil_ga_train_step_adamw: the step's last launch finishes
    # the gradients, applies AdamW and re-packs the weights).  Single-GPU runs only: a data-parallel reduction sits between the
    # gradients and the update.  Protocol: in_step_args() -> the call -> in_step_done() (or in_step_abort() if the call refused).
    def can_run_in_step(self) -> bool:
        return self.peer is None and not self._frozen and self.guard_flag is not None

    def in_step_args(self, track_flag: bool = False) -> tuple:
        g = self.param_groups[0]
        if self._in_step is not None:
            raise RuntimeError("acmil_amd.FlatAdamW: in_step_args() without in_step_done()")
        track = bool(track_flag)
        if track and len(self._pending) >= 12:
            raise RuntimeError("acmil_amd.FlatAdamW: poll_skipped() must be called while steps are tracked")
        self.step_count += 1
        self._step_id += 1
        self._launches += 1
        slot = self._step_id % 16
        self._in_step = (track, slot)
        b1, b2 = g["betas"]
        return (self.flat, self.exp_avg, self.exp_avg_sq, float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                self._launches, self._skipped_dev, self._host_flags.data_ptr() + 4 * slot if track else None)

    def in_step_abort(self):
        self._in_step = None
        self.step_count -= 1
        self._step_id -= 1
        self._launches -= 1

    def in_step_done(self) -> int:
        track, slot = self._in_step
        self._in_step = None
        self.mutations += 1
        if self.on_step is not None:
            self.on_step()
        if track:
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((self._step_id, ev, slot))
        return self._step_id

    @torch.no_grad()
    def step(self, track_flag: bool = False) -> int:
        """One launch.  track_flag: the launch also stores the range flag into pinned host memory so that `poll_skipped` can tell
        later whether the device applied this step.  Returns this call's id."""
        g = self.param_groups[0]
        self.step_count += 1
        self._step_id += 1
        self._launches += 1
        b1, b2 = g["betas"]
        lib = _lib.load()
        kept = [(o, self.flat[o:o + n].clone()) for o, n in self._frozen]
        track = track_flag and self.guard_flag is not None
        if track and len(self._pending) >= 12:
            raise RuntimeError("acmil_amd.FlatAdamW: poll_skipped() must be called while steps are tracked")
        slot = self._step_id % 16
        if self.peer is not None and not self.peer.verified:
            self._first_peer_step(lib, g, b1, b2, track, slot)
        elif self.peer is not None:
            # direct data-parallel reduction: bucket -> own slot + flags at the peers, then wait + reduce (rank order) + AdamW in ONE launch
            self._peer_launch(lib, g, b1, b2, track, slot, None)
        hooked = False
        if self.peer is None:
            if self.pack_hook is not None and not self._frozen:
                hooked = self.pack_hook.launch((self.flat, self.exp_avg, self.exp_avg_sq, float(g["lr"]), float(b1), float(b2), float(g["eps"]),
                                                float(g["weight_decay"]), self._launches, self._skipped_dev,
                                                self._host_flags.data_ptr() + 4 * slot if track else None), self.guard_flag)
            if not hooked:
                self._plain_launch(lib, g, b1, b2, track, slot)
        for o, v in kept:
            self.flat[o:o + v.numel()].copy_(v)
        self.mutations += 1
        if self.on_step is not None:      # the update bypasses torch's version counters: owners of derived caches are told
            self.on_step()
        if hooked:
            self.pack_hook.committed()
        if track:
            ev = torch.cuda.Event()
            ev.record()
            self._pending.append((self._step_id, ev, slot))
        return self._step_id

    def _plain_launch(self, lib, g, b1, b2, track, slot):
        # tracked: the launch itself stores the flag into pinned host memory (no copy on the stream)
        rc = lib.acmil_adamw_step_report(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                         self.numel, float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                         self._launches, None if self.guard_flag is None else self.guard_flag.data_ptr(),
                                         self._skipped_dev.data_ptr(), self._host_flags.data_ptr() + 4 * slot if track else None,
                                         torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "acmil_adamw_step_report")

    def _peer_launch(self, lib, g, b1, b2, track, slot, reduced_out):
        self.peer.publish(self.grad)
        rc = lib.acmil_adamw_step_peer(self.flat.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(), self.numel,
                                       self.peer.slot_ptrs(), self.peer.my_flags_ptr, self.peer.world, self.peer.rank,
                                       self.peer.step_id, float(self.peer.timeout_s), self.peer.err.data_ptr(), float(g["lr"]),
                                       float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]), self._launches,
                                       0 if self.guard_flag is None else 1, self._skipped_dev.data_ptr(),
                                       self._host_flags.data_ptr() + 4 * slot if track else None,
                                       None if reduced_out is None else reduced_out.data_ptr(), torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "acmil_adamw_step_peer")

    def _first_peer_step(self, lib, g, b1, b2, track, slot):
        """The first step of the direct reduction, CHECKED (collective; one synchronisation): the bucket as the fused launch reduced
        it must equal one torch.distributed all_reduce of the same bucket.  A mismatch means the peers' slots are not read coherently
        on this machine (the path has only ever run with two ranks on one GPU): the step is undone, redone on the collective's result,
        and every rank drops to torch.distributed for good -- loudly, never silently diverging."""
        import torch.distributed as dist
        peer = self.peer
        keep = (self.flat.clone(), self.exp_avg.clone(), self.exp_avg_sq.clone(), self._skipped_dev.clone())
        # reference = one all_reduce of the bucket; beside it the all_reduce of |bucket|: the two summation orders (rank order here,
        # ring / tree there) may differ by rounding relative to the ADDENDS -- an element whose rank contributions cancel has no
        # small error relative to its own value or to the bucket maximum -- so the bound is world * eps * sum_r |g_r|
        if dist.get_backend(peer.group) == "nccl":
            ref = self.grad.clone()
            mag = self.grad.abs()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=peer.group)
            dist.all_reduce(mag, op=dist.ReduceOp.SUM, group=peer.group)
        else:                                                       # (gloo control plane of the single-GPU test: reduce on the host)
            ref = self.grad.cpu()
            mag = self.grad.abs().cpu()
            dist.all_reduce(ref, op=dist.ReduceOp.SUM, group=peer.group)
            dist.all_reduce(mag, op=dist.ReduceOp.SUM, group=peer.group)
            ref = ref.to(self.grad.device)
            mag = mag.to(self.grad.device)
        ref.div_(peer.world)
        mag.div_(peer.world)
        reduced = torch.zeros_like(self.grad)
        self._peer_launch(lib, g, b1, b2, track, slot, reduced)
        timed_out = int(peer.err.item()) != 0                      # (synchronises)
        if getattr(peer, "_selfcheck_perturb", False):             # test hook (tests/dist_worker_peer.py): what a stale remote line would look like
            reduced[0] += 1.0
        n = self.numel
        scale = float(ref[:n].abs().max().item())
        bound = (peer.world + 1) * 1.2e-7 * mag[:n] + 1e-30
        ok = (not timed_out) and bool(torch.isfinite(reduced[:n]).all()) and bool(((reduced[:n] - ref[:n]).abs() <= bound).all())
        if not peer._agree(1 if ok else 0, self.flat.device, peer.group):
            print("acmil_amd.FlatAdamW[rank %d]: direct gradient reduction disagrees with torch.distributed on its first step (%s); "
                  "falling back to all_reduce" % (peer.rank, "timeout" if timed_out else "max |d| %.3e of %.3e" % (
                      float((reduced - ref).abs().max().item()) if not timed_out else float("nan"), scale)))
            self.flat.copy_(keep[0]); self.exp_avg.copy_(keep[1]); self.exp_avg_sq.copy_(keep[2]); self._skipped_dev.copy_(keep[3])
            self.grad.copy_(ref)
            peer.verdict = "mismatch: fell back to torch.distributed"
            peer.owner = None                                       # train.GradBucket.allreduce_mean takes over from the next step on
            self.peer = None                                        # (step() now issues the plain launch on the collective's result)
            return
        peer.verified = True
        peer.verdict = "first step equals all_reduce (max |d| %.1e)" % float((reduced - ref).abs().max().item())

    def poll_skipped(self, lag: int = 2) -> List[int]:
        """Ids of tracked steps the device did NOT apply (range flag set), looking only at steps at least `lag` calls old -- their
        work has long finished, so the wait never idles the GPU and every rank of a data-parallel job decides on the same step.
        lag = 0 waits for everything outstanding.  `step_count` (what state_dict reports) is put back for every skipped step; the
        kernel's own bias-correction step number never counted it (it subtracts a device-side count of skipped launches)."""
        out = []
        while self._pending and self._pending[0][0] <= self._step_id - lag:
            sid, ev, slot = self._pending.popleft()
            ev.synchronize()
            f = float(self._host_flags[slot])
            if f < 0.0:            # acmil_adamw_step_peer: a peer's gradient flag never arrived, the launch changed nothing
                raise RuntimeError("acmil_amd.FlatAdamW: direct gradient reduction timed out at step %d (a peer rank did not publish "
                                   "within %.0f s); no update was applied" % (sid, self.peer.timeout_s if self.peer is not None else 0.0))
            if f != 0.0:
                out.append(sid)
                self.step_count -= 1
                self.skipped_steps += 1
        return out

    def state_dict(self):
        """torch.optim.AdamW's layout (what the reference's checkpoints hold under 'optimizer', utils/utils.py:415-422): per-parameter
        'state' entries {step, exp_avg, exp_avg_sq} and 'param_groups' with parameter indices -- loadable by torch.optim.AdamW."""
        # tracked steps nobody has polled yet: wait for their flags and leave out the ones the device skipped (the entries stay
        # queued for poll_skipped -- the trainer still has to repeat those bags)
        step_count = self.step_count
        for sid, ev, slot in self._pending:
            ev.synchronize()
            if float(self._host_flags[slot]) != 0.0:
                step_count -= 1
        state, off = {}, 0
        frozen = {o for o, _ in self._frozen}
        for i, p in enumerate(self.params):
            n = p.numel()
            if off not in frozen and step_count > 0:
                state[i] = {"step": torch.tensor(float(step_count)),
                            "exp_avg": self.exp_avg[off:off + n].view_as(p).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view_as(p).clone()}
            off += n
        g = self.param_groups[0]
        group = {"lr": g["lr"], "betas": tuple(g["betas"]), "eps": g["eps"], "weight_decay": g["weight_decay"], "amsgrad": False,
                 "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
                 "decoupled_weight_decay": True, "params": list(range(len(self.params)))}
        return {"state": state, "param_groups": [group]}

    def load_state_dict(self, sd):
        """Accepts torch.optim.AdamW's layout (also one written by the reference) or this class's earlier flat layout."""
        self._pending.clear()      # flags of steps taken before the load belong to the discarded state
        self.mutations += 1
        self.skipped_steps = 0
        if "state" in sd:
            off, steps = 0, []
            for i, p in enumerate(self.params):
                n = p.numel()
                st = sd["state"].get(i, sd["state"].get(str(i)))
                if st is not None:
                    self.exp_avg[off:off + n].copy_(st["exp_avg"].reshape(-1))
                    self.exp_avg_sq[off:off + n].copy_(st["exp_avg_sq"].reshape(-1))
                    steps.append(int(float(st["step"])))
                else:
                    self.exp_avg[off:off + n].zero_(); self.exp_avg_sq[off:off + n].zero_()
                off += n
            self.step_count = max(steps) if steps else 0
            self._launches = self.step_count; self._skipped_dev.zero_()
            for g, s_ in zip(self.param_groups, sd["param_groups"]):
                g.update({k: v for k, v in s_.items() if k in ("lr", "betas", "eps", "weight_decay")})
            return
        self.step_count = int(sd["step"])
        self._launches = self.step_count; self._skipped_dev.zero_()
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s_ in zip(self.param_groups, sd["param_groups"]):
            g.update(s_)
