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
from typing import Callable, Iterable, Optional

import torch

from . import _lib


class FlatAdamW:
    def __init__(self, params: Iterable[torch.nn.Parameter], lr: float = 1e-3, betas=(0.9, 0.999), eps: float = 1e-8,
                 weight_decay: float = 1e-2, grad_buffer: Optional[torch.Tensor] = None, on_step: Optional[Callable[[], None]] = None):
        self.params = [p for p in params if p.requires_grad]
        if not self.params or not all(p.is_cuda and p.dtype == torch.float32 for p in self.params):
            raise RuntimeError("acmil_amd.FlatAdamW: CUDA fp32 parameters only")
        dev = self.params[0].device
        self.numel = sum(p.numel() for p in self.params)
        self.flat = torch.empty(self.numel, dtype=torch.float32, device=dev)
        self.grad = grad_buffer if grad_buffer is not None else torch.zeros(self.numel, dtype=torch.float32, device=dev)
        if self.grad.numel() != self.numel:
            raise RuntimeError("acmil_amd.FlatAdamW: grad_buffer size mismatch")
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

    def zero_grad(self, set_to_none: bool = False):
        self.grad.zero_()

    @torch.no_grad()
    def step(self):
        g = self.param_groups[0]
        self.step_count += 1
        b1, b2 = g["betas"]
        lib = _lib.load()
        rc = lib.acmil_adamw_step(self.flat.data_ptr(), self.grad.data_ptr(), self.exp_avg.data_ptr(), self.exp_avg_sq.data_ptr(),
                                  self.numel, float(g["lr"]), float(b1), float(b2), float(g["eps"]), float(g["weight_decay"]),
                                  1.0 - b1 ** self.step_count, 1.0 - b2 ** self.step_count, torch.cuda.current_stream().cuda_stream)
        _lib.check(rc, "acmil_adamw_step")
        if self.on_step is not None:      # the update bypasses torch's version counters: owners of derived caches are told
            self.on_step()

    def state_dict(self):
        return {"step": self.step_count, "exp_avg": self.exp_avg.clone(), "exp_avg_sq": self.exp_avg_sq.clone(),
                "param_groups": [{k: v for k, v in g.items() if k != "params"} for g in self.param_groups]}

    def load_state_dict(self, sd):
        self.step_count = int(sd["step"])
        self.exp_avg.copy_(sd["exp_avg"]); self.exp_avg_sq.copy_(sd["exp_avg_sq"])
        for g, s in zip(self.param_groups, sd["param_groups"]):
            g.update(s)
