"""Gated-attention aggregators with the reference's interface, computed by libacmil_hip.so.

Mirrors `architecture/transformer.py` of dazhangyu123/ACMIL: `Attention_Gated` (:239-267),
`ABMIL` (:270-286), `ACMIL_GA` (:291-352) -- same constructor arguments, `forward` /
`forward_feature` signatures and return shapes, same `state_dict()` keys, masking keyed on
`self.training`.  The arithmetic is NOT torch: forward (and backward) call the HIP kernels through
`acmil_amd.ops`; CPU tensors are rejected (no fallback).

Extra, non-reference knobs (keyword-only, defaults keep reference behaviour):
  precision   'f16x3' (default: fp32-parity split-f16 MFMA), 'fp32' (exact fp32 MFMA) or 'f16' (throughput)
  range_guard True (default): the split-f16 kernels flag bag values / projected features outside the f16 range (|v| >= 65504,
              inf, NaN) in a device status word; the module reads it back after the forward (one 4-byte copy) and redoes
              that bag with exact fp32 MFMA arithmetic, so a result is never silently inf / garbage where the reference's
              fp32 result is finite.  False skips the read-back (no synchronisation; the status is still in `_last`).
  and `forward(x, uniforms=...)` to inject the STKIM `torch.rand(K,k)` draw for reproducible tests.
"""
from __future__ import annotations

from typing import Optional

import math

import torch
import torch.nn as nn
import torch.nn.functional as F

from .. import autograd as AG
from .. import ops
from .network import Classifier_1fc, DimReduction


class Attention_Gated(nn.Module):
    """Parameter container with the reference's layout (attention_V.0 / attention_U.0 / attention_weights)."""

    def __init__(self, L=512, D=128, K=1):
        super().__init__()
        # D = 128 (the reference's default, transformer.py:240,292) is the width of the fused kernels; any other width runs the
        # aggregators' op-by-op path (_GatedBase._forward_generic: acmil_gated_scores takes any D)
        self.L, self.D, self.K = L, D, K
        self.attention_V = nn.Sequential(nn.Linear(L, D), nn.Tanh())
        self.attention_U = nn.Sequential(nn.Linear(L, D), nn.Sigmoid())
        self.attention_weights = nn.Linear(D, K)

    def forward(self, x):  # standalone use only (not on the fused hot path): plain library GEMMs
        a = self.attention_weights(self.attention_V(x) * self.attention_U(x))
        return torch.transpose(a, 1, 0)


class _GaTrainFn(torch.autograd.Function):
    """Training step of the aggregator as one autograd node: HIP score pass + STKIM + masked pooling forward,
    HIP backward (acmil_ga_backward).  Differentiable outputs: sub_preds, slide_pred, A_out."""

    @staticmethod
    def forward(ctx, module, xb, uniforms, n_params, *params):
        packed, dims = module._packed()
        out = module._masked_forward(xb, packed, dims, uniforms, want_afeat=True,
                                     masking=getattr(module, "_masking_now", True))
        ctx.dims = out.get("dims_bwd", dims)
        ctx.has_slide = "slide_pred" in out
        ctx.save_for_backward(xb, out["h"], out["A_out"], out["afeat"], *[p.detach() for p in params])
        module._last = out
        slide = out["slide_pred"].unsqueeze(0) if ctx.has_slide else out["sub_preds"].new_zeros(1, dims.C)
        return out["sub_preds"], slide, out["A_out"].unsqueeze(0)

    @staticmethod
    def backward(ctx, d_sub, d_slide, d_attn):
        xb, h, A_out, afeat, *params = ctx.saved_tensors
        dims = ctx.dims
        if d_sub is None:
            d_sub = torch.zeros(dims.K, dims.C, device=xb.device)
        if ctx.has_slide and d_slide is None:
            d_slide = torch.zeros(dims.C, device=xb.device)
        grads = ops.ga_backward(xb, h, A_out, afeat, params, dims, d_sub, d_slide if ctx.has_slide else None, d_attn)
        return (None, None, None, None, *grads)


class RangeTicket:
    """The split-f16 range word of ONE launch, on its way to the host: an asynchronous 4-byte copy into pinned memory issued
    right behind the launch (stream-ordered after it and before whatever overwrites the word) plus an event.  `int(ticket)`
    waits for THAT event only -- work enqueued after the launch keeps the GPU busy meanwhile (a plain `int(status)` is a
    stream-ordered read-back: it would wait for everything enqueued since).  (Measured alternative, dropped: routing the copy
    through a side stream costs more in cross-queue waits than the ~50 us the copy engine hand-off costs on the compute stream.)"""
    _pool: list = []          # free one-word views of pinned slabs (a fresh pin_memory() call costs up to tens of ms: never in a loop)

    @classmethod
    def _take(cls) -> torch.Tensor:
        if not cls._pool:
            slab = torch.empty(64, dtype=torch.int32).pin_memory()
            cls._pool.extend(slab[i:i + 1] for i in range(64))
        return cls._pool.pop()

    def __init__(self, status: torch.Tensor):
        self.host = RangeTicket._take()
        self.host.copy_(status, non_blocking=True)
        self.event = torch.cuda.Event()
        self.event.record(torch.cuda.current_stream(status.device))
        self._value: Optional[int] = None

    def __int__(self) -> int:
        if self._value is None:
            self.event.synchronize()
            self._value = int(self.host[0])
            RangeTicket._pool.append(self.host)
            self.host = None
        return self._value

    def __del__(self):          # a ticket nobody read: its word goes back once the copy has certainly landed
        try:
            host = getattr(self, "host", None)
            if host is not None and self.event.query():
                RangeTicket._pool.append(host)
        except Exception:      # interpreter shutdown
            pass


class _GatedBase(nn.Module):
    """Shared plumbing: parameter gathering, the packed-weight cache, the two-pass training forward."""

    precision: str
    range_guard = True
    n_masked_patch = 0
    mask_drop = 0.0

    def _out_of_range(self, status) -> bool:
        """Read the device status word of a split-f16 launch (synchronises); True = redo the bag in fp32 mode."""
        if not self.range_guard or status is None:
            return False
        bad = int(status) != 0
        if bad:
            self._fb_host = getattr(self, "_fb_host", 0) + 1
        return bad

    # Fallback counter = repeats decided on the host (paths that read the status word) + repeats decided ON THE DEVICE
    # (acmil_ga_forward_guarded counts in a device word; reading the property synchronises, the forward never does).
    def _fb_counter(self, device) -> torch.Tensor:
        c = self.__dict__.get("_fb_dev")
        if c is None or c.device != device:
            c = torch.zeros(1, dtype=torch.int32, device=device)
            self.__dict__["_fb_dev"] = c
        return c

    @property
    def range_fallbacks(self) -> int:
        c = self.__dict__.get("_fb_dev")
        return getattr(self, "_fb_host", 0) + (int(c) if c is not None else 0)

    @range_fallbacks.setter
    def range_fallbacks(self, v: int) -> None:
        c = self.__dict__.get("_fb_dev")
        if c is not None:
            c.zero_()
        self._fb_host = int(v)

    def _heads(self):
        raise NotImplementedError

    def _check_dropout(self):
        """(kept for callers of earlier rounds: nothing is refused any more -- train-mode classifier dropout runs the op-by-op path)"""

    def _generic(self) -> bool:
        """True where the reference's constructor arguments leave what the fused kernels are built for: an attention hidden width other
        than 128 (`D`, transformer.py:240,270,292), classifier dropout in TRAINING mode (`droprate`, network.py:10-19: identity in
        eval mode, where the fused path serves such a model as before), DimReduction residual blocks.  Those run _forward_generic."""
        return (self.attention.D != ops.GA_DA or (getattr(self, "droprate", 0) != 0 and self.training)
                or getattr(self.dimreduction, "numRes", 0) > 0)

    def _forward_generic(self, xb, uniforms=None, masking=False):
        """The aggregator op by op, every O(N) operation a kernel of libacmil_hip.so with a HIP backward (acmil_amd.autograd): projection
        (+ residual blocks) as split-f16 / exact-fp32 MFMA GEMMs, acmil_gated_scores' arithmetic at ANY attention width, STKIM selection
        (acmil_stkim_select) + differentiable mask fill, row softmax, pooling GEMM; the heads (K + 1 products of [1, D_inner] by
        [D_inner, C]) as exact-fp32 GEMMs behind torch's dropout mask (`Classifier_1fc`, network.py:14-19: the mask is drawn by torch's
        generator, as the reference's is).  Differentiable: forward(), train_step and forward_feature of such a model all come here.
        Returns (sub_preds [K, C], slide_pred [1, C] or None, A [K, N] masked raw scores, afeat [K, D_inner], topk, midx)."""
        x = xb if xb.dtype == torch.float32 else xb.float()
        a = self.attention
        prec = "fp32" if self.precision == "fp32" else "f16x3"
        wc, bc, ws, bs = self._heads()

        def run(prec):
            h = AG.linear(x, self.dimreduction.fc1.weight, None, relu=True, precision=prec)
            for blk in getattr(self.dimreduction, "resBlocks", []):
                t = AG.linear(h, blk.block[0].weight, None, relu=True, precision=prec)
                h = h + AG.linear(t, blk.block[2].weight, None, relu=True, precision=prec)
            A = AG.gated_scores(h, a.attention_V[0].weight, a.attention_V[0].bias, a.attention_U[0].weight, a.attention_U[0].bias,
                                a.attention_weights.weight, a.attention_weights.bias, prec)
            return h, A

        h, A = run(prec)
        if prec == "f16x3" and self.range_guard:
            # the split-f16 products have no status word on this path: one reduction over h and the bag (as the composed path's generic-GEMM route)
            hmax, xmax = h.detach().max(), x.abs().max()
            if not bool(torch.isfinite(hmax) & (hmax < 65504.0) & torch.isfinite(xmax) & (xmax < 65504.0)):
                self._fb_host = getattr(self, "_fb_host", 0) + 1
                h, A = run("fp32")
        n = x.shape[0]
        topk = midx = None
        k = min(self.n_masked_patch, n) if masking else 0
        m = int(k * self.mask_drop)
        if k > 0:
            topk, midx = ops.stkim_select(A.detach().contiguous(), k, m, uniforms, rng=None if uniforms is not None else self._next_rng())
            if m > 0:
                A = AG.mask_fill(A, midx)
        afeat = AG.attn_pool(h, A)                                   # softmax over N, then P h: [K, D_inner]
        drop = getattr(self, "droprate", 0)
        outs = []
        for i in range(len(wc)):
            v = afeat[i:i + 1]
            if drop != 0:
                v = F.dropout(v, drop, self.training)
            outs.append(AG.linear(v, wc[i], bc[i], precision="fp32"))
        sub = torch.cat(outs, 0)
        slide = None
        if ws is not None:
            # bag_A = softmax(A).mean(0); bag_feat = bag_A h  ==  mean_k afeat_k  (transformer.py:328-330; linear in the softmax rows)
            bf = afeat.mean(0, keepdim=True)
            if drop != 0:
                bf = F.dropout(bf, drop, self.training)
            slide = AG.linear(bf, ws, bs, precision="fp32")
        return sub, slide, A, afeat, topk, midx

    def _raw_params(self):
        wc, bc, ws, bs = self._heads()
        a = self.attention
        return [self.dimreduction.fc1.weight, a.attention_V[0].weight, a.attention_V[0].bias,
                a.attention_U[0].weight, a.attention_U[0].bias, a.attention_weights.weight,
                a.attention_weights.bias], wc, bc, ws, bs

    def _param_list(self):
        base, wc, bc, ws, bs = self._raw_params()
        return base + list(wc) + list(bc) + ([ws, bs] if ws is not None else [])

    @staticmethod
    def _param_key(allp):
        """Identity of the current parameter VALUES: storage pointers + torch's version counters (one pass, shared by the caches)."""
        return tuple([p.data_ptr() for p in allp] + [p._version for p in allp])

    def _pack_now(self, precision: str):
        base, wc, bc, ws, bs = self._raw_params()
        return ops.ga_pack_weights(*[p.detach() for p in base], [p.detach() for p in wc], [p.detach() for p in bc],
                                   None if ws is None else ws.detach(), None if bs is None else bs.detach(), precision)

    def _packed(self, precision: Optional[str] = None, key=None):
        """Packed fragment stream of the current parameter values; re-packed only when a parameter changed
        (tracked through tensor version counters / storage pointers).  precision: pack for another arithmetic mode than the
        module's (the fp32 re-run of the range guard); not cached.  key: a _param_key the caller has already formed."""
        if precision is not None and precision != self.precision:
            return self._pack_now(precision)
        if key is None:
            key = self._param_key(self._param_list())
        cache = getattr(self, "_pack_cache", None)
        if cache is None or cache[0] != key or cache[3] != self.precision:
            packed, dims = self._pack_now(self.precision)
            cache = (key, packed, dims, self.precision)
            self._pack_cache = cache
        return cache[1], cache[2]

    def _packed_cached(self, precision: str, key=None):
        """As _packed(precision), but cached per mode (the fp32 repeats of the range guard: eval loops hit it per flagged batch)."""
        if precision == self.precision:
            return self._packed(key=key)
        if key is None:
            key = self._param_key(self._param_list())
        cache = self.__dict__.setdefault("_pack_cache_alt", {})
        hit = cache.get(precision)
        if hit is None or hit[0] != key:
            packed, dims = self._pack_now(precision)
            hit = cache[precision] = (key, packed, dims)
        return hit[1], hit[2]

    def invalidate_packed(self):
        """Drop the packed-weight caches (for optimizers that update the parameters outside torch's version counters)."""
        self._pack_cache = None
        self._step_pack_key = None
        self._w1_cache = None
        self._gate_cache = None
        self.__dict__.pop("_pack_cache_alt", None)

    # D_inner with a fully fused forward kernel in every arithmetic mode AND a one-call training step (csrc/ga_families.inc)
    FUSED_D_INNER = (128, 256)
    # Wide families of the reference's newer feature extractors (Step3_WSI_classification_ACMIL.py:78-87: CLIP-L 768/384, UNI
    # 1024/512): fully fused in split-f16 arithmetic (csrc/ga_forward_kernel_v3.h, one 512-register wave per SIMD; round 5) -- eval
    # forward and the score pass of a training step.  Their fp32 mode, GigaPath (1536/768) and n_token > 5 take the composed path.
    FUSED_WIDE_D_INNER = (384, 512)

    supports_in_step_optimizer = True      # train_step(optimizer=...): see there

    FUSED_MAX_TOKENS = 5      # n_token above (Step3_WSI_classification_ACMIL.py:39 takes any) runs the composed kernels, K <= 16

    def _is_fused(self) -> bool:
        return (self.dimreduction.fc1.weight.shape[0] in self.FUSED_D_INNER
                and self.attention.attention_weights.weight.shape[0] <= self.FUSED_MAX_TOKENS)

    def _is_wide_fused(self) -> bool:
        return (self.precision == "f16x3" and self.dimreduction.fc1.weight.shape[0] in self.FUSED_WIDE_D_INNER
                and self.attention.attention_weights.weight.shape[0] <= self.FUSED_MAX_TOKENS
                and self.attention.attention_V[0].weight.shape[0] == 128)

    def _score_pass(self, xb, packed, dims, device_guard=False):
        """Raw scores A [K,N] and h [N,Di].  Fused widths: one kernel (GEMM chain in registers).  Other widths: the projection
        as a split-f16 MFMA GEMM with the ReLU epilogue (acmil_gemm_f16x3; network.py:49-57), then acmil_gated_scores
        (transformer.py:259-267) on h in HBM -- same arithmetic class, h makes one round trip."""
        if self._is_fused():
            if self.precision != "f16x3":
                return ops.ga_scores(xb, packed, dims, self.precision)
            A, h, status = ops.ga_scores(xb, packed, dims, self.precision, with_status=True)
            if self._out_of_range(status):
                p32, d32 = self._packed("fp32")
                self._bwd_dims = d32               # the backward of this step follows in exact fp32 as well
                return ops.ga_scores(xb, p32, d32, "fp32")
            return A, h
        if self._is_wide_fused():
            # one kernel as at the fused widths (h leaves the GEMM1 accumulators once, as fp32 rows; the gate runs on registers)
            A, h, status = ops.ga_scores(xb, packed, dims, "f16x3", with_status=True)
            if not self._out_of_range(status):
                return A, h
            self._bwd_dims = ops.GaDims(dims.D, dims.Di, dims.K, dims.C, dims.has_bag_head, mode=ops.mode_id("fp32"))
            return self._score_pass_composed(xb, dims, "fp32")
        return self._score_pass_composed(xb, dims, "fp32" if self.precision == "fp32" else "f16x3", packed if device_guard else None)

    def _score_pass_composed(self, xb, dims, prec, guard_packed=None):
        """guard_packed (eval forward only): the module's packed buffer -- the split-f16 range guard is then resolved ON THE DEVICE
        (ops.ga_rescore_fp32_cond overwrites h and A with their exact-fp32 values iff the projection's status word is set) instead of
        by a mid-forward read-back; a training step passes None: its backward has to know the arithmetic on the host."""
        base = self._raw_params()[0]

        status = None

        def project(prec):
            nonlocal status
            if prec == "f16x3" and xb.stride(0) * xb.element_size() % 16 == 0:
                # packed-weight Linear kernel (csrc/linear_kernel.h): takes fp32 / fp16 / bf16 bags as they are; ~20 % faster than the
                # generic split GEMM at these shapes (K >= 768, 256-wide output chunks); leaves the range word of the fused kernels
                h, status = ops.linear_f16x3(xb, self._packed_w1(), base[0].shape[0], relu=True, want_status=True)
                return h
            x32 = xb if xb.dtype == torch.float32 else xb.float()      # storage-format conversion of a 16-bit bag
            return ops.gemm(x32, base[0].detach(), trans_b=True, act=1, precision=prec)

        h = project(prec)
        if (prec == "f16x3" and self.range_guard and guard_packed is not None and status is not None and h.shape[1] % 16 == 0
                and self.attention.attention_V[0].weight.shape[0] == 128):
            pvu, bvu = self._packed_gate()
            A = ops.gated_scores_packed(h, pvu, bvu, base[5], base[6])
            w1 = base[0].detach()
            ops.ga_rescore_fp32_cond(xb, guard_packed, w1 if w1.is_contiguous() else w1.contiguous(), dims, h, A, status,
                                     self._fb_counter(xb.device))
            return A, h
        if prec == "f16x3" and self.range_guard:
            # Same rule as the fused kernel's status word, on the same quantity: the projected features (a bag value outside the f16
            # range has an inf hi half and makes its patch's features inf / NaN, so the one test covers both).  The projection kernel
            # leaves the word; the generic-GEMM route (odd row strides) pays one reduction over h instead.
            if status is not None:
                bad = int(status) != 0
            else:
                # the generic GEMM's ReLU epilogue is fmaxf(v, 0): a NaN pre-activation (bag value >= 65504: hi = inf, lo = -inf) or a
                # -inf one becomes 0 and max(h) stays finite -- so this route tests the BAG as well, one pass over x
                hmax = h.max()
                xabs = xb.abs().max()
                bad = not bool(torch.isfinite(hmax) & (hmax < 65504.0) & torch.isfinite(xabs) & (xabs < 65504.0))
            if bad:
                self._fb_host = getattr(self, "_fb_host", 0) + 1
                prec = "fp32"
                self._bwd_dims = ops.GaDims(dims.D, dims.Di, dims.K, dims.C, dims.has_bag_head, mode=ops.mode_id("fp32"))
                h = project(prec)
        if prec == "f16x3" and h.shape[1] % 16 == 0:
            # one pass over h: [Wv; Wu] product with the gate formed in the accumulators (csrc/linear_kernel.h, act = 2)
            pvu, bvu = self._packed_gate()
            A = ops.gated_scores_packed(h, pvu, bvu, base[5], base[6])
        else:
            A = ops.gated_scores(h, *[p.detach() for p in base[1:7]], precision=prec)
        return A, h

    def _packed_gate(self):
        """Fragment stream + bias of the interleaved [Wv; Wu] matrix for acmil_gated_scores_packed, re-packed when a parameter changed."""
        a = self.attention
        ps = [a.attention_V[0].weight, a.attention_V[0].bias, a.attention_U[0].weight, a.attention_U[0].bias]
        key = tuple((p.data_ptr(), p._version) for p in ps)
        cache = getattr(self, "_gate_cache", None)
        if cache is None or cache[0] != key:
            cache = (key,) + ops.pack_gate(*[p.detach() for p in ps])
            self._gate_cache = cache
        return cache[1], cache[2]

    def _packed_w1(self):
        """Fragment stream of dimreduction.fc1.weight for acmil_linear_f16x3, re-packed when the parameter changed."""
        w = self.dimreduction.fc1.weight
        key = (w.data_ptr(), w._version)
        cache = getattr(self, "_w1_cache", None)
        if cache is None or cache[0] != key:
            cache = (key, ops.linear_pack(w.detach().contiguous()))
            self._w1_cache = cache
        return cache[1]

    def _eval_forward(self, xb, packed, dims, want_scores=True, want_preds=True, want_bag_feat=False, key=None):
        """Unmasked forward: the fully fused kernel where a family exists, else score pass + pooling pass."""
        if self._is_fused() and self.precision == "f16x3" and self.range_guard:
            # one library call, no host read-back: split-f16 launch, then its exact-fp32 repeat predicated ON THE DEVICE on the range
            # status (it exits at once for an in-range bag), then merge + heads -- the loop `for x: model(x)` stays asynchronous
            p32, _ = self._packed_cached("fp32", key=key)
            out = ops.ga_forward_guarded([xb], packed, p32, dims, self._fb_counter(xb.device), want_scores=want_scores,
                                         want_preds=want_preds, want_bag_feat=want_bag_feat)
            for k in ("A_out", "sub_preds", "slide_pred", "bag_feat"):
                if k in out:
                    out[k] = out[k][0]
            return out
        if self._is_fused():
            out = ops.ga_forward(xb, packed, dims, self.precision, want_scores=want_scores, want_preds=want_preds,
                                 want_bag_feat=want_bag_feat)
            if self.precision == "f16x3" and self._out_of_range(out["range_status"]):
                p32, d32 = self._packed("fp32")
                out = ops.ga_forward(xb, p32, d32, "fp32", want_scores=want_scores, want_preds=want_preds,
                                     want_bag_feat=want_bag_feat)
                out["range_fallback"] = True
            return out
        if self._is_wide_fused():
            # ONE fused launch + merge + heads, and -- round 5 -- NO host read-back: the exact-fp32 repeat (widen, projection, gated
            # scores, pooling partials, op by op) is enqueued behind the fused launch with every kernel predicated ON THE DEVICE on its
            # range status; merge + heads finish whichever result is there (acmil_ga_forward_guarded_wide)
            if self.range_guard:
                w1 = self.dimreduction.fc1.weight.detach()
                return ops.ga_forward_guarded_wide(xb, packed, w1 if w1.is_contiguous() else w1.contiguous(), dims, self._fb_counter(xb.device),
                                                   want_scores=want_scores, want_preds=want_preds, want_bag_feat=want_bag_feat)
            return ops.ga_forward(xb, packed, dims, "f16x3", want_scores=want_scores, want_preds=want_preds, want_bag_feat=want_bag_feat)
        out = self._masked_forward(xb, packed, dims, None, want_bag_feat=want_bag_feat, masking=False, device_guard=True)
        out.pop("h", None)
        return out

    def _masked_forward(self, xb, packed, dims, uniforms, want_bag_feat=False, want_afeat=False, masking=True, device_guard=False):
        """score pass (keeps h) -> STKIM selection -> masked pooling.  masking=False: no mask (plain training forward).
        device_guard (eval forward of the composed path): no host read-back of the range status, see _score_pass_composed."""
        n = xb.shape[0]
        self._bwd_dims = dims
        A, h = self._score_pass(xb, packed, dims, device_guard=device_guard)
        k = min(self.n_masked_patch, n) if masking else 0
        m = int(k * self.mask_drop)
        topk = midx = None
        if k > 0:
            # no injected draw: the STKIM kernel draws its uniforms itself (Philox keyed on this module's seed and forward count)
            topk, midx = ops.stkim_select(A, k, m, uniforms, rng=None if uniforms is not None else self._next_rng())
        out = ops.ga_pool(h, A, packed, dims, self.precision, midx if m > 0 else None, want_bag_feat=want_bag_feat,
                          want_afeat=want_afeat)
        out["topk_idx"], out["masked_idx"], out["h"] = topk, midx, h
        out["dims_bwd"] = self._bwd_dims
        return out

    def _next_rng(self):
        """(seed, offset) of the next device-side STKIM draw: the seed is torch's at first use (so torch.manual_seed governs it), the
        offset counts this module's masked forwards.  Replaces the `torch.rand(K, k)` launch of transformer.py:316."""
        if getattr(self, "_rng_seed", None) is None:
            self._rng_seed = int(torch.initial_seed()) & (2 ** 63 - 1)
            self._rng_count = 0
        self._rng_count += 1
        return (self._rng_seed, self._rng_count)

    def _all_params(self):
        base, wc, bc, ws, bs = self._raw_params()
        return base + list(wc) + list(bc) + ([ws, bs] if ws is not None else [])

    @staticmethod
    def _bag(x: torch.Tensor) -> torch.Tensor:
        if x.dim() != 3:
            raise RuntimeError("expected x of shape [1, N, D_feat]")
        x = x[0]  # B must be 1 (the reference silently takes x[0], transformer.py:306)
        return x if x.is_contiguous() else x.contiguous()


class ABMIL(_GatedBase):
    def __init__(self, conf, D=128, droprate=0, *, precision="f16x3", range_guard=True):
        super().__init__()
        self.droprate = droprate      # Classifier_1fc's dropout (network.py:10-16): identity in eval mode; training with it takes the op-by-op path
        self.dimreduction = DimReduction(conf.D_feat, conf.D_inner)
        self.attention = Attention_Gated(conf.D_inner, D, 1)
        self.classifier = Classifier_1fc(conf.D_inner, conf.n_class, droprate)
        self.precision = precision
        self.range_guard = range_guard

    def _heads(self):
        return [self.classifier.fc.weight], [self.classifier.fc.bias], None, None

    def forward(self, x):  # x: [1, N, D_feat] -> logits [1, C]   (transformer.py:277-286)
        xb = self._bag(x)
        if self._generic():
            return self._forward_generic(xb)[0]
        params = self._all_params()
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            self._masking_now = False
            return _GaTrainFn.apply(self, xb, None, len(params), *params)[0]
        key = self._param_key(params)
        packed, dims = self._packed(key=key)
        out = self._eval_forward(xb, packed, dims, want_scores=False, key=key)
        return out["sub_preds"]


class ACMIL_GA(_GatedBase):
    def __init__(self, conf, D=128, droprate=0, n_token=1, n_masked_patch=0, mask_drop=0, *, precision="f16x3", range_guard=True):
        super().__init__()
        self.range_guard = range_guard
        self.droprate = droprate      # see ABMIL
        self.dimreduction = DimReduction(conf.D_feat, conf.D_inner)
        self.attention = Attention_Gated(conf.D_inner, D, n_token)
        self.classifier = nn.ModuleList()
        for _ in range(n_token):
            self.classifier.append(Classifier_1fc(conf.D_inner, conf.n_class, droprate))
        self.n_masked_patch = n_masked_patch
        self.n_token = conf.n_token
        self.Slide_classifier = Classifier_1fc(conf.D_inner, conf.n_class, droprate)
        self.mask_drop = mask_drop
        self.precision = precision

    def _heads(self):
        return ([c.fc.weight for c in self.classifier], [c.fc.bias for c in self.classifier],
                self.Slide_classifier.fc.weight, self.Slide_classifier.fc.bias)

    def forward(self, x, uniforms: Optional[torch.Tensor] = None):
        """x [1,N,D_feat] -> (sub_preds [K,C], slide_pred [1,C], A_out [1,K,N])  (transformer.py:305-330)."""
        xb = self._bag(x)
        params = self._all_params()
        masking = self.n_masked_patch > 0 and self.training
        if self._generic():
            sub, slide, A, afeat, topk, midx = self._forward_generic(xb, uniforms, masking)
            self._last = {"sub_preds": sub, "slide_pred": slide, "A_out": A, "afeat": afeat, "topk_idx": topk, "masked_idx": midx}
            return sub, slide, A.unsqueeze(0)
        if torch.is_grad_enabled() and any(p.requires_grad for p in params):
            self._masking_now = masking
            return _GaTrainFn.apply(self, xb, uniforms, len(params), *params)
        key = self._param_key(params)          # once per call: both packed-weight caches compare against it
        packed, dims = self._packed(key=key)
        if masking:
            out = self._masked_forward(xb, packed, dims, uniforms)
        else:
            out = self._eval_forward(xb, packed, dims, key=key)
        self._last = out
        return out["sub_preds"], out["slide_pred"].unsqueeze(0), out["A_out"].unsqueeze(0)

    @torch.no_grad()
    def train_step(self, x, label, uniforms: Optional[torch.Tensor] = None, guard_flag: Optional[torch.Tensor] = None,
                   precision: Optional[str] = None, optimizer=None, track_flag: bool = False, in_step: bool = True):
        """One training step WITHOUT autograd: HIP forward (score pass, STKIM, masked pooling), the ACMIL loss and the HIP
        backward, writing the gradients into `p.grad` (allocated on first use, overwritten) -- ONE library call
        (acmil_ga_train_step) at the fused widths D_inner 128 / 256, the stand-alone kernels op by op at 384 / 512 / 768.
        Same mathematics as `loss = diff + loss0 + loss1; loss.backward()` of the reference's train_one_epoch
        (Step3_WSI_classification_ACMIL.py:200-219); the optimiser step stays with the caller.
        x [1,N,D_feat], label [1] int64 on the GPU.  Returns (losses [4] = loss0, loss1, diff_loss, total, on device; outputs dict).
        Range guard of the split-f16 arithmetic: by default the step's status word is read after the call (one
        synchronisation) and a flagged step is repeated in fp32 before it returns.  guard_flag (a device float, e.g.
        FlatAdamW.guard_flag): no read-back at all -- the step leaves 1.0 / 0.0 there for the optimizer launch to act on
        (it skips a flagged step; train.train_one_epoch repeats the bag in fp32 two steps later).  precision overrides the
        module's arithmetic for this call ("fp32": the repeat).
        optimizer (a FlatAdamW over exactly this module's parameters, single-GPU runs): where the one-call step can, it applies the
        update ITSELF -- its last launch finishes the gradients, runs AdamW and re-packs the weights (acmil_ga_train_step_adamw: three
        launches fewer per step) -- and the outputs carry `opt_step_id` (what optimizer.step(track_flag) would have returned); where it
        cannot (fp32 arithmetic, no guard_flag, frozen parameters, a direct peer reduction, the wide / composed families),
        `opt_step_id` is None and the caller steps the optimizer as before.  in_step=False (data-parallel runs: the gradient
        all-reduce sits between this call and the update) only tells the step whose updates to trust: with `adamw_pack_hook`
        installed the optimizer's own launch re-packs the weights (acmil_ga_adamw_pack) and this call skips its pack launch."""
        xb = self._bag(x)
        params = self._all_params()
        masking = self.n_masked_patch > 0 and self.training
        k_top = min(self.n_masked_patch, xb.shape[0]) if masking else 0
        for p in params:
            if p.grad is None:
                p.grad = torch.empty_like(p)
        if self._generic():
            if guard_flag is not None:
                guard_flag.zero_()          # the op-by-op path resolves the guard itself, before it returns
            losses, out = self._train_step_generic(xb, label, uniforms, params, masking)
            self._last = out
            return losses, out
        # no injected draw: the kernel draws on the device; ONE (seed, offset) per step, so an fp32 re-run masks the same patches
        self._step_rng = self._next_rng() if (masking and uniforms is None) else None
        if self._is_fused() and getattr(self, "fused_step", True):
            losses, out = self._train_step_fused(xb, label, uniforms, params, k_top, guard_flag, precision, optimizer, track_flag, in_step)
        else:
            if precision is not None and precision != self.precision:
                raise NotImplementedError("acmil_amd: per-call precision is a feature of the one-call step")
            if guard_flag is not None:
                guard_flag.zero_()          # the op-by-op step resolves the guard itself, before it returns
            losses, out = self._train_step_composed(xb, label, uniforms, params, masking)
        self._last = out
        return losses, out

    def _train_step_fused(self, xb, label, uniforms, params, k_top, guard_flag=None, precision=None, optimizer=None, track_flag=False, in_step=True,
                          rows=None):
        """The whole step enqueued by one library call (csrc/ga_step.hip: 8 launches).  The packed weights are rebuilt inside
        the call every step (the parameters change between steps); the range status of the split-f16 score pass is read once,
        after the call -- a flagged step is repeated in fp32 arithmetic before anybody sees its gradients.
        rows (train_step_batch): xb holds a GROUP of bags back to back, rows = their patch counts, label [G] -- one step on the mean
        gradient of the group (acmil_ga_train_step_group); its exact-fp32 form runs bag by bag (_train_step_group_fp32)."""
        dev = xb.device
        m_mask = int(k_top * self.mask_drop)
        grads = [p.grad for p in params]
        if label.dtype != torch.int64:
            label = label.to(torch.int64)

        def step_call(precision, st, **kw):
            if rows is None:
                return ops.ga_train_step(xb, st[0], st[1], precision, params, grads, label, uniforms, k_top, m_mask, **kw)
            return ops.ga_train_step_group(xb, rows, st[0], st[1], precision, params, grads, label, uniforms, k_top, m_mask, **kw)

        def run(precision):
            if rows is not None and precision == "fp32":
                return self._train_step_group_fp32(xb, rows, label, uniforms, params, grads, k_top, guard_flag)
            cache = self.__dict__.setdefault("_step_packed", {})
            st = cache.get((precision, dev))
            if st is None:
                packed, dims = self._packed(precision) if precision != self.precision else self._packed()
                st = cache[(precision, dev)] = (packed.clone(), dims)      # a private buffer: the call rewrites it every step
            # the step's private packed buffer stays current as long as nothing but this optimizer's fused launches (the in-step closing
            # launch, acmil_ga_adamw_pack through adamw_pack_hook) touches the parameters: (storages, versions, the optimizer's count)
            opt_ok = (optimizer is not None and precision == "f16x3" and hasattr(optimizer, "mutations")
                      and not self.__dict__.get("_opt_in_step_refused"))
            key = (precision, dev, id(optimizer), self._param_key(params)) if opt_ok else None
            valid = opt_ok and self.__dict__.get("_step_pack_key") == key + (optimizer.mutations,)
            # (never inside a multi-rank job: the update would run on this rank's local gradients, ahead of the all-reduce)
            use_in_step = (opt_ok and in_step and guard_flag is not None and optimizer.can_run_in_step()
                           and not (torch.distributed.is_available() and torch.distributed.is_initialized() and torch.distributed.get_world_size() > 1))
            if use_in_step and not self._in_step_supported(optimizer, params, grads, st[1]):
                use_in_step = False
            if not use_in_step:
                out = step_call(precision, st, repack=not valid, guard_flag=guard_flag, rng=self._step_rng)
                if opt_ok:
                    self._step_pack_key = key + (optimizer.mutations,)      # packed now holds exactly the values this step ran on
                out["opt_step_id"] = None
                return out
            # (whether the closing launch can serve this parameter set was asked BEFORE anything is enqueued -- _in_step_supported,
            # acmil_ga_adamw_supported -- so an error from the call itself is a real failure and propagates: ADVICE r5)
            args = optimizer.in_step_args(track_flag)
            try:
                out = step_call(precision, st, repack=not valid, guard_flag=guard_flag, rng=self._step_rng, adamw=args)
            except Exception:
                optimizer.in_step_abort()
                raise
            out["opt_step_id"] = optimizer.in_step_done()             # (calls invalidate_packed through on_step: the eval caches are stale now)
            self._step_pack_key = key + (optimizer.mutations,)
            return out

        prec = precision or self.precision
        out = run(prec)
        if guard_flag is None and prec == "f16x3" and self._out_of_range(out["range_status"]):
            out = run("fp32")
            out["range_fallback"] = True
        return out["losses"], out

    def _in_step_supported(self, optimizer, params, grads, dims) -> bool:
        """Can the step's closing launch (finish + AdamW + re-pack, csrc/ga_opt_step.hip) serve this parameter set?  A side-effect-free
        query of the library (alignment of the three matrices / gradients / moments, the parameters tiling the optimizer's flat
        buffer), remembered per set of storages; a refusal is permanent for that set and nothing has been launched."""
        key = (id(optimizer), tuple(p.data_ptr() for p in params), tuple(g.data_ptr() for g in grads))
        hit = self.__dict__.get("_in_step_ok")
        if hit is None or hit[0] != key:
            ok = ops.ga_adamw_supported(dims, params, grads, optimizer.flat, optimizer.exp_avg, optimizer.exp_avg_sq)
            hit = self.__dict__["_in_step_ok"] = (key, ok)
            if not ok:
                self._opt_in_step_refused = True
        return hit[1]

    def _train_step_group_fp32(self, xb, rows, labels, uniforms, params, grads, k_top, guard_flag):
        """The exact-fp32 form of a group step (the repeat of a range-flagged group; precision='fp32' models): the one-call step bag
        by bag, gradients averaged -- same mathematics as acmil_ga_train_step_group, op for op the reference's arithmetic."""
        G = len(rows)
        acc = [torch.zeros_like(g) for g in grads]
        outs, off = [], 0
        cache = self.__dict__.setdefault("_step_packed", {})
        st = cache.get(("fp32", xb.device))
        if st is None:
            packed, dims = self._packed("fp32") if self.precision != "fp32" else self._packed()
            st = cache[("fp32", xb.device)] = (packed.clone(), dims)
        m_mask = int(k_top * self.mask_drop)
        for b, n in enumerate(rows):
            o = ops.ga_train_step(xb[off:off + n], st[0], st[1], "fp32", params, grads, labels[b:b + 1],
                                  None if uniforms is None else uniforms[b], k_top, m_mask, repack=(b == 0), guard_flag=guard_flag,
                                  rng=None if self._step_rng is None else (self._step_rng[0], self._step_rng[1] * 65536 + b))
            torch._foreach_add_(acc, grads, alpha=1.0 / G)
            outs.append(o)
            off += n
        torch._foreach_copy_(grads, acc)
        offs = [0]
        for n in rows:
            offs.append(offs[-1] + n)
        has_slide = outs[0]["slide_pred"] is not None
        return {"losses": torch.stack([o["losses"] for o in outs]), "sub_preds": torch.stack([o["sub_preds"] for o in outs]),
                "slide_pred": torch.stack([o["slide_pred"] for o in outs]) if has_slide else None,
                "A_out": torch.cat([o["A_out"] for o in outs], dim=1), "offsets": offs,
                "topk_idx": torch.stack([o["topk_idx"] for o in outs]) if k_top > 0 else None,
                "masked_idx": torch.stack([o["masked_idx"] for o in outs]) if m_mask > 0 else None,
                "range_status": outs[-1]["range_status"], "opt_step_id": None}

    @torch.no_grad()
    def train_step_batch(self, bags, labels, uniforms: Optional[torch.Tensor] = None, guard_flag: Optional[torch.Tensor] = None,
                         precision: Optional[str] = None, optimizer=None, track_flag: bool = False, in_step: bool = True):
        """ONE training step on a GROUP of G <= 16 slides: the parameter gradients are the MEAN of the slides' per-slide gradients --
        exactly what G data-parallel ranks compute per step, on one GPU, and under data parallelism `bags per rank`
        (one all-reduce per G slides).  Not in the reference, whose loop is B = 1 (Step3_WSI_classification_ACMIL.py:189-221);
        per slide the forward (STKIM included), the three losses and the backward are the same as `train_step`.
        bags: a list of [N_b, D_feat] CUDA tensors of one dtype (concatenated here: one copy), or a pair (x [sum N_b, D_feat], rows)
        with the bags' rows already back to back (staging.staged_train_groups delivers that: no copy).  labels [G] int64 on the GPU;
        uniforms [G, K, k] to inject the STKIM draws (a list of per-bag [K, k_b] tensors where a bag is smaller than n_masked_patch).
        Everything else as `train_step`.  Returns (losses [G, 4], outputs dict:
        sub_preds [G,K,C], slide_pred [G,C], A_out [K, sum N_b] with `offsets`, topk_idx / masked_idx [G,K,.] bag-local)."""
        if isinstance(bags, tuple) and len(bags) == 2 and torch.is_tensor(bags[0]):
            xb, rows = bags[0], [int(r) for r in bags[1]]
        else:
            bags = [b[0] if b.dim() == 3 else b for b in bags]
            rows = [int(b.shape[0]) for b in bags]
            if any(b.dtype != bags[0].dtype for b in bags):
                bags = [b.float() for b in bags]
            xb = bags[0] if len(bags) == 1 else torch.cat(bags, dim=0)
        if not xb.is_contiguous():
            xb = xb.contiguous()
        G = len(rows)
        if not 1 <= G <= ops.MAX_GROUP:
            raise RuntimeError("acmil_amd: train_step_batch takes 1 .. %d bags per step" % ops.MAX_GROUP)
        params = self._all_params()
        masking = self.n_masked_patch > 0 and self.training
        k_top = min(self.n_masked_patch, min(rows)) if masking else 0
        for p in params:
            if p.grad is None:
                p.grad = torch.empty_like(p)
        fused = self._is_fused() and getattr(self, "fused_step", True) and not self._generic()
        # bags smaller than n_masked_patch clamp k per bag (transformer.py:313) and the wide / composed families have no group kernel:
        # those groups run bag by bag with the gradients averaged -- the same mathematics
        if not fused or (masking and min(rows) < self.n_masked_patch):
            return self._train_step_group_serial(xb, rows, labels, uniforms, params, guard_flag, precision)
        self._step_rng = self._next_rng() if (masking and uniforms is None) else None
        losses, out = self._train_step_fused(xb, labels, uniforms, params, k_top, guard_flag, precision, optimizer, track_flag, in_step, rows=rows)
        self._last = out
        return losses, out

    def _train_step_group_serial(self, xb, rows, labels, uniforms, params, guard_flag, precision):
        """A group step as G single-slide steps with averaged gradients (families / bag sizes the group kernels do not take)."""
        G = len(rows)
        grads = [p.grad for p in params]
        acc = [torch.zeros_like(g) for g in grads]
        outs, ls, off = [], [], 0
        flagged = None
        for b, n in enumerate(rows):
            l, o = self.train_step(xb[off:off + n].unsqueeze(0), labels[b:b + 1], None if uniforms is None else uniforms[b], guard_flag=guard_flag,
                                   precision=precision)
            if guard_flag is not None:
                flagged = guard_flag.clone() if flagged is None else torch.maximum(flagged, guard_flag)
            torch._foreach_add_(acc, [p.grad for p in params], alpha=1.0 / G)
            outs.append(o); ls.append(l)
            off += n
        torch._foreach_copy_(grads, acc)
        for p, g in zip(params, grads):
            p.grad = g
        if guard_flag is not None and flagged is not None:
            guard_flag.copy_(flagged)
        out = {"per_bag": outs, "opt_step_id": None}
        self._last = out
        return torch.stack(ls), out

    def adamw_pack_hook(self, optimizer):
        """For FlatAdamW.pack_hook: lets the optimizer's launch re-pack this module's weights (acmil_ga_adamw_pack) so that the next
        training step needs no pack launch.  Used where the update cannot ride in the step's own closing launch (data parallel)."""
        return _AdamwPackHook(self, optimizer)

    def _train_step_generic(self, xb, label, uniforms, params, masking):
        """train_step of a model on the op-by-op path (_generic): differentiable forward, the fused loss kernel (acmil_ga_loss: the three
        losses and their gradients w.r.t. the forward's outputs), then torch.autograd through the HIP backward kernels."""
        with torch.enable_grad():
            sub, slide, A, afeat, topk, midx = self._forward_generic(xb, uniforms, masking)
        losses, d_sub, d_slide, d_A = ops.ga_loss(sub.detach().contiguous(), None if slide is None else slide.detach().reshape(-1).contiguous(),
                                                  A.detach().contiguous(), label)
        for p in params:
            p.grad.zero_()
        outs, gs = [sub, A], [d_sub, d_A]
        if slide is not None:
            outs.append(slide); gs.append(d_slide.view_as(slide))
        torch.autograd.backward(outs, gs)
        return losses, {"sub_preds": sub.detach(), "slide_pred": None if slide is None else slide.detach().reshape(-1), "A_out": A.detach(),
                        "afeat": afeat.detach(), "topk_idx": topk, "masked_idx": midx, "opt_step_id": None}

    def _train_step_composed(self, xb, label, uniforms, params, masking):
        """Op-by-op step (the wide D_inner families, and the fused step's cross-check in the tests)."""
        packed, dims = self._packed()

        def run():
            out = self._masked_forward(xb, packed, dims, uniforms, want_afeat=True, masking=masking)
            losses, d_sub, d_slide, d_A = ops.ga_loss(out["sub_preds"], out.get("slide_pred"), out["A_out"], label)
            ops.ga_backward(xb, out["h"], out["A_out"], out["afeat"], [p.detach() for p in params], out.get("dims_bwd", dims), d_sub,
                            d_slide, d_A, grads_out=[p.grad for p in params])
            return losses, out

        return run()       # (the score pass handles the split-f16 range guard itself: fp32 re-run + fp32 backward)

    @torch.no_grad()
    def forward_batch(self, bags, defer_guard: bool = False, precision: Optional[str] = None):
        """Eval forward of up to 64 bags (list of [N_b, D_feat] CUDA tensors, ragged N allowed) in ONE fused launch
        (acmil_ga_forward_batch).  Returns a list of the reference's per-slide triples
        (sub_preds [K,C], slide_pred [1,C], A_out [1,K,N_b]).  Not in the reference (it is strictly B=1); same maths per bag.
        defer_guard=True: no host read-back here -- returns (triples, status) where `status` is a RangeTicket for this launch's
        split-f16 range word (None when the arithmetic has none); the caller looks at it later (`int(status) != 0` = redo these
        bags with precision="fp32"), e.g. after it has enqueued the next batch, so the GPU never idles on the check
        (train.evaluate does that).  precision overrides the module's arithmetic for this call."""
        if len(bags) > ops.MAX_BATCH and not defer_guard:      # one launch takes 64 bags: longer lists go out in groups
            out = []
            for i in range(0, len(bags), ops.MAX_BATCH):
                out += self.forward_batch(bags[i:i + ops.MAX_BATCH], precision=precision)
            return out
        prec = precision or self.precision
        if self._generic():
            triples = []
            for b in bags:
                sub, slide, A, *_ = self._forward_generic(b if b.is_contiguous() else b.contiguous())
                triples.append((sub, slide, A.unsqueeze(0)))
            return (triples, None) if defer_guard else triples
        packed, dims = self._packed() if prec == self.precision else self._packed_cached(prec)
        bags = [b if b.is_contiguous() else b.contiguous() for b in bags]
        if any(b.dtype != bags[0].dtype for b in bags):      # one launch reads one storage format: widen (exactly) to fp32
            bags = [b.float() for b in bags]
        wide = self._is_wide_fused() and prec == "f16x3"
        if not self._is_fused() and not wide:
            save = self.precision
            try:
                self.precision = prec           # (an fp32 repeat of a wide-family batch: the composed path in that arithmetic)
                outs = [self._eval_forward(b, packed, dims) for b in bags]
            finally:
                self.precision = save
            triples = [(o["sub_preds"], o["slide_pred"].unsqueeze(0), o["A_out"].unsqueeze(0)) for o in outs]
            return (triples, None) if defer_guard else triples
        out = ops.ga_forward_batch(bags, packed, dims, prec)
        status = None
        if prec == "f16x3" and self.range_guard:
            if defer_guard:
                status = RangeTicket(out["range_status"])      # async copy now: the next launch on this workspace overwrites the word
            elif self._out_of_range(out["range_status"]):      # some bag left the f16 range: redo in fp32
                if wide:
                    return self.forward_batch(bags, precision="fp32")
                p32, d32 = self._packed_cached("fp32")
                out = ops.ga_forward_batch(bags, p32, d32, "fp32")
        triples = [(out["sub_preds"][i], out["slide_pred"][i].unsqueeze(0), out["A_out"][i].unsqueeze(0)) for i in range(len(bags))]
        return (triples, status) if defer_guard else triples

    def _is_composed_groupable(self, prec: str) -> bool:
        """The composed families (D_inner = 768, n_token > 5; attention width 128) in split-f16 arithmetic: their three kernels take the
        rows of several bags at once (forward_group)."""
        return (prec == "f16x3" and not self._generic() and not self._is_fused() and not self._is_wide_fused()
                and self.dimreduction.fc1.weight.shape[0] % 16 == 0 and self.attention.attention_V[0].weight.shape[0] == 128)

    def forward_group(self, x, rows, defer_guard: bool = False):
        """Eval forward of G <= 16 bags whose rows lie BACK TO BACK in x [sum N_b, D_feat] (staging.staged_train_groups delivers a group
        that way: no device-side concatenation) -- the batched eval of the COMPOSED families (GigaPath 1536 -> 768, n_token > 5), whose
        kernels are per patch: ONE projection launch and ONE gated-score launch over all rows (a single 50 000-patch slide fills the
        256 persistent workgroups of the projection for 2.3 rounds of (tile, chunk) units: three rounds' time; a group has no such
        tail), then per-bag pooling tiles and merge + heads for all bags (acmil_ga_pool_group).  Per bag the same mathematics and
        kernels as `model(x)` (transformer.py:305-330); returns the per-slide triples of forward_batch.  Not in the reference (B = 1).
        Range guard: ONE status word per group (the projection's); defer_guard=True returns (triples, RangeTicket) as forward_batch
        does, otherwise a flagged group is repeated bag by bag in fp32 here.  Other families: the bags go through forward_batch."""
        rows = [int(r) for r in rows]
        xb = x[0] if x.dim() == 3 else x
        if not xb.is_contiguous():
            xb = xb.contiguous()
        if xb.shape[0] != sum(rows) or not rows:
            raise RuntimeError("acmil_amd: forward_group takes x [sum rows, D_feat]")
        offs = [0]
        for r in rows:
            offs.append(offs[-1] + r)
        views = [xb[offs[i]:offs[i + 1]] for i in range(len(rows))]
        if not self._is_composed_groupable(self.precision) or xb.stride(0) * xb.element_size() % 16 != 0:
            return self.forward_batch(views, defer_guard=defer_guard)
        triples, status = [], None
        for g0 in range(0, len(rows), ops.MAX_GROUP):
            g1 = min(len(rows), g0 + ops.MAX_GROUP)
            t, st = self._forward_group_composed(xb[offs[g0]:offs[g1]], rows[g0:g1])
            triples += t
            if st is not None and self.range_guard:
                if defer_guard and g1 == len(rows) and g0 == 0:
                    status = RangeTicket(st)
                elif self._out_of_range(st):
                    triples[g0:g1] = self.forward_batch(views[g0:g1], precision="fp32")
        return (triples, status) if defer_guard else triples

    def _forward_group_composed(self, xb, rows):
        packed, dims = self._packed()
        base = self._raw_params()[0]
        h, status = ops.linear_f16x3(xb, self._packed_w1(), base[0].shape[0], relu=True, want_status=True)
        pvu, bvu = self._packed_gate()
        A = ops.gated_scores_packed(h, pvu, bvu, base[5], base[6])
        out = ops.ga_pool_group(h, A, rows, packed, dims, "f16x3")
        offs = [0]
        for r in rows:
            offs.append(offs[-1] + r)
        slide = out.get("slide_pred")
        triples = [(out["sub_preds"][i], slide[i].unsqueeze(0) if slide is not None else None, A[:, offs[i]:offs[i + 1]].unsqueeze(0))
                   for i in range(len(rows))]
        return triples, status

    def forward_feature(self, x, use_attention_mask=False, uniforms: Optional[torch.Tensor] = None):
        """x [1,N,D_feat] -> bag_feat [1,Di]  (transformer.py:332-352)."""
        xb = self._bag(x)
        if self._generic():
            afeat = self._forward_generic(xb, uniforms, masking=self.n_masked_patch > 0 and use_attention_mask)[3]
            return afeat.mean(0, keepdim=True)
        packed, dims = self._packed()
        if self.n_masked_patch > 0 and use_attention_mask:
            out = self._masked_forward(xb, packed, dims, uniforms, want_bag_feat=True)
        else:
            out = self._eval_forward(xb, packed, dims, want_scores=False, want_preds=False, want_bag_feat=True)
        return out["bag_feat"].unsqueeze(0)


class _AdamwPackHook:
    """FlatAdamW.pack_hook of one ACMIL_GA module (see ACMIL_GA.adamw_pack_hook)."""

    def __init__(self, model, optimizer):
        import weakref
        self.model = weakref.ref(model)
        self.optimizer = weakref.ref(optimizer)
        self._pending = None

    def launch(self, adamw_args, skip_flag) -> bool:
        m, opt = self.model(), self.optimizer()
        self._pending = None
        if m is None or opt is None or m.precision != "f16x3" or not m._is_fused() or m.__dict__.get("_opt_in_step_refused"):
            return False
        params = m._all_params()
        dev = params[0].device
        st = m.__dict__.get("_step_packed", {}).get(("f16x3", dev))
        if st is None or any(p.grad is None for p in params):
            return False
        key = ("f16x3", dev, id(opt), m._param_key(params))
        was_valid = m.__dict__.get("_step_pack_key") == key + (opt.mutations,)
        grads = [p.grad for p in params]
        if not m._in_step_supported(opt, params, grads, st[1]):      # asked before anything is launched; any error below is a real one
            return False
        ops.ga_adamw_pack(st[0], st[1], params, grads, adamw_args, skip_flag)
        # an applied launch rewrites every packed copy; a launch the device skips (range flag) leaves the buffer as it was
        self._pending = key if (was_valid or skip_flag is None) else None
        return True

    def committed(self):
        m, opt = self.model(), self.optimizer()
        if m is not None and opt is not None and self._pending is not None:
            m._step_pack_key = self._pending + (opt.mutations,)
        self._pending = None


class MutiHeadAttention(nn.Module):
    """The reference's attention layer (architecture/transformer.py:107-185), same names and constructor: q/k/v/out projections (to
    `embedding_dim // downsample_rate`), `num_heads` heads, LayerNorm(eps=1e-6), Dropout.  Inside ACMIL_MHA / MHA (8 heads, no
    down-sampling, one query per branch) it is a parameter container whose arithmetic lives in acmil_mha_forward (csrc/mha.hip) and in
    ACMIL_MHA._forward_train.  Stand-alone, `forward(q [1,Nq,E], k [1,N,E], v [1,N,E]) -> (out [Nq,E], attn [H,Nq,N])` runs the same
    FOLDED form for any head count / down-sampling rate and a handful of queries: with few queries the key projection folds into the
    query (score = k . (Wk_h^T q'_h) / sqrt(c) + const) and the value projection folds through the pooling (out = Wv_h (sum_n P v_n) + bv),
    so the O(N) work is one [Nq H, E] x [E, N] score GEMM, a row softmax and one pooling GEMM (acmil_amd.autograd: HIP forward + backward)."""

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1, dropout: float = 0.1,
                 n_masked_patch: int = 0, mask_drop: float = 0.0):
        super().__init__()
        self.n_masked_patch, self.mask_drop = n_masked_patch, mask_drop
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        if self.internal_dim % num_heads != 0:
            raise AssertionError("num_heads must divide embedding_dim.")
        self.q_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.k_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)
        self.layer_norm = nn.LayerNorm(embedding_dim, eps=1e-6)
        self.dropout = nn.Dropout(dropout)

    def forward(self, q, k, v):
        if q.dim() != 3 or q.shape[0] != 1 or k.shape[0] != 1 or v.shape[0] != 1:
            raise RuntimeError("acmil_amd: MutiHeadAttention.forward expects q [1,Nq,E], k / v [1,N,E]")
        if not k.is_cuda:
            raise RuntimeError("acmil_amd: MutiHeadAttention runs on an MI355X only (no CPU fallback)")
        H, ci, E = self.num_heads, self.internal_dim, self.embedding_dim
        c = ci // H
        q0, k0, v0 = q[0].float(), k[0].float().contiguous(), v[0].float().contiguous()
        nq, n = q0.shape[0], k0.shape[0]
        qp = F.linear(q0, self.q_proj.weight, self.q_proj.bias).view(nq, H, c)                            # q' per query and head
        rows = torch.einsum("hce,qhc->hqe", self.k_proj.weight.view(H, c, E), qp) / math.sqrt(c)       # [H, Nq, E]
        cst = torch.einsum("hc,qhc->hq", self.k_proj.bias.view(H, c), qp) / math.sqrt(c)
        S = AG.matmul(rows.reshape(H * nq, E), k0, trans_b=True) + cst.reshape(H * nq, 1)               # [H Nq, N] = the returned attn
        masked = S
        if self.n_masked_patch > 0 and self.training:                                                   # transformer.py:162-171
            kk = min(self.n_masked_patch, n)
            drop = int(kk * self.mask_drop)
            if drop > 0:
                u = torch.rand(S.shape[0], kk, device=S.device)
                _, midx = ops.stkim_select(S.detach().contiguous(), kk, drop, u)
                masked = AG.mask_fill(S, midx)
        P = AG.softmax_rows(masked.contiguous())
        pooled = AG.matmul(P, v0).view(H, nq, E)
        out1 = torch.einsum("hce,hqe->qhc", self.v_proj.weight.view(H, c, E), pooled) + self.v_proj.bias.view(1, H, c)
        o = F.linear(out1.reshape(nq, ci), self.out_proj.weight, self.out_proj.bias)
        o = self.layer_norm(self.dropout(o))
        return o, masked.view(H, nq, n)


class MutiHeadAttention_modify(nn.Module):
    """Parameter container of the bag attention (architecture/transformer.py:187-219): v / out projections + LayerNorm."""

    def __init__(self, embedding_dim: int, num_heads: int, downsample_rate: int = 1, dropout: float = 0.1):
        super().__init__()
        self.embedding_dim = embedding_dim
        self.internal_dim = embedding_dim // downsample_rate
        self.num_heads = num_heads
        if self.internal_dim % num_heads != 0:
            raise AssertionError("num_heads must divide embedding_dim.")
        self.v_proj = nn.Linear(embedding_dim, self.internal_dim)
        self.out_proj = nn.Linear(self.internal_dim, embedding_dim)
        self.layer_norm = nn.LayerNorm(embedding_dim, eps=1e-6)
        self.dropout = nn.Dropout(dropout)

    def forward(self, v, attn):
        """v [1,N,E], attn [H,1,N] (already normalised: transformer.py:221-236) -> [1,E]: the value projection folded through the
        weighted sum, as in ACMIL_MHA's bag head."""
        if not v.is_cuda:
            raise RuntimeError("acmil_amd: MutiHeadAttention_modify runs on an MI355X only (no CPU fallback)")
        H, ci, E = self.num_heads, self.internal_dim, self.embedding_dim
        c = ci // H
        pooled = AG.matmul(attn.reshape(H, -1).float().contiguous(), v[0].float().contiguous())          # [H, E]
        out1 = (torch.einsum("hce,he->hc", self.v_proj.weight.view(H, c, E), pooled) + self.v_proj.bias.view(H, c)).reshape(1, ci)
        o = F.linear(out1, self.out_proj.weight, self.out_proj.bias)
        return self.layer_norm(self.dropout(o))


class ACMIL_MHA(nn.Module):
    """Drop-in for the reference's `ACMIL_MHA` (architecture/transformer.py:49-83, `--arch mha`): same ctor, parameter names
    and return contract `(sub_preds [K,C], slide_pred [1,C], attns [8,K,N])`.  Under torch.no_grad() in eval mode: one C call
    (acmil_mha_forward).  With gradients enabled: the same folded mathematics through acmil_amd.autograd (trainable; the
    reference's Dropout(0.1) after out_proj and the per-row top-k mask-drop are applied in train mode)."""

    def __init__(self, conf, n_token=1, n_masked_patch=0, mask_drop=0, *, precision="f16x3"):
        super().__init__()
        self.dimreduction = DimReduction(conf.D_feat, conf.D_inner)
        self.sub_attention = nn.ModuleList(
            [MutiHeadAttention(conf.D_inner, 8, n_masked_patch=n_masked_patch, mask_drop=mask_drop) for _ in range(n_token)])
        self.bag_attention = MutiHeadAttention_modify(conf.D_inner, 8)
        self.q = nn.Parameter(torch.zeros((1, n_token, conf.D_inner)))
        nn.init.normal_(self.q, std=1e-6)
        self.n_class = conf.n_class
        self.classifier = nn.ModuleList([Classifier_1fc(conf.D_inner, conf.n_class, 0.0) for _ in range(n_token)])
        self.n_token = n_token
        self.Slide_classifier = Classifier_1fc(conf.D_inner, conf.n_class, 0.0)
        self.precision = precision

    def _post(self, att, out1):
        """out_proj, dropout, LayerNorm(1e-6) of one attention module on the [Di] pooled projection (transformer.py:179-183)"""
        o = F.linear(out1, att.out_proj.weight, att.out_proj.bias)
        o = F.dropout(o, 0.1, self.training)
        return F.layer_norm(o, (o.shape[-1],), att.layer_norm.weight, att.layer_norm.bias, 1e-6)

    def _forward_train(self, x):
        """Differentiable forward in the single-query folded form (see csrc/mha.hip): the O(N) passes -- projection, score
        GEMM, row softmax, pooling GEMM -- are acmil_amd.autograd Functions with HIP backward kernels; the per-branch Di x Di
        glue, the top-k mask bookkeeping and dropout are torch ops on small tensors."""
        K, H = self.n_token, 8
        h = AG.linear(x, self.dimreduction.fc1.weight, None, relu=True, precision=self.precision)     # [N, Di]
        n, di = h.shape
        c = di // H
        rows, csts = [], []
        for i, att in enumerate(self.sub_attention):
            qp = F.linear(self.q[0, i], att.q_proj.weight, att.q_proj.bias).view(H, c, 1)          # q' per head
            rows.append((att.k_proj.weight.view(H, c, di) * qp).sum(1) / math.sqrt(c))            # [H, Di]
            csts.append((att.k_proj.bias.view(H, c) * qp[..., 0]).sum(1) / math.sqrt(c))           # [H]
        MT, cst = torch.stack(rows, 1).reshape(H * K, di), torch.stack(csts, 1).reshape(H * K)    # row = j*K + i
        S = AG.matmul(MT, h, trans_b=True) + cst[:, None]                                          # [H*K, N] = attns
        attns = S.view(H, K, n)
        masked = S
        k_mask = min(self.sub_attention[0].n_masked_patch, n) if self.training else 0
        if k_mask > 0:                                                                            # transformer.py:162-171, per (head, branch) row
            drop = int(k_mask * self.sub_attention[0].mask_drop)
            if drop > 0:
                # top-k + random subset with the STKIM kernels of the GA path (one row per (head, branch)), then the mask as a
                # differentiable index fill: no [8K, N] topk / ones / scatter / masked_fill tensors
                u = torch.rand(S.shape[0], k_mask, device=S.device)
                _, midx = ops.stkim_select(S.detach().contiguous(), k_mask, drop, u)
                masked = AG.mask_fill(S, midx)
                attns = masked.view(H, K, n)
        P = AG.softmax_rows(masked.contiguous())
        pooled = AG.matmul(P, h).view(H, K, di)                                                    # sum_n P h
        outs = []
        for i, att in enumerate(self.sub_attention):
            out1 = ((att.v_proj.weight.view(H, c, di) * pooled[:, i].unsqueeze(1)).sum(-1) + att.v_proj.bias.view(H, c)).reshape(1, di)
            head = self.classifier[i] if isinstance(self.classifier, nn.ModuleList) else self.classifier
            outs.append(head(self._post(att, out1)))
        bag = self.bag_attention
        if bag is None:                                                                            # (MHA: one branch, no bag head)
            return torch.cat(outs, 0), None, attns
        pb = pooled.mean(1)                                                                        # mean_i softmax(.) is linear in P
        out1 = ((bag.v_proj.weight.view(H, c, di) * pb.unsqueeze(1)).sum(-1) + bag.v_proj.bias.view(H, c)).reshape(1, di)
        return torch.cat(outs, 0), self.Slide_classifier(self._post(bag, out1)), attns

    def forward(self, input):
        if input.dim() != 3 or input.shape[0] != 1:
            raise RuntimeError("acmil_amd: ACMIL_MHA expects input [1, N, D_feat]")
        x = input[0]
        if not x.is_cuda:
            raise RuntimeError("acmil_amd: ACMIL_MHA runs on an MI355X only (no CPU fallback)")
        x = x.float().contiguous()
        if torch.is_grad_enabled() and (self.training or any(p.requires_grad for p in self.parameters())):
            return self._forward_train(x)
        if self.training:
            raise NotImplementedError("acmil_amd: train-mode ACMIL_MHA draws dropout / mask-drop randomness; enable gradients or use .eval()")
        sd = {k: v.detach() for k, v in self.state_dict(keep_vars=True).items()}
        out = ops.mha_forward(x, sd, self.n_token, self.n_class, self.precision)
        return out["sub_preds"], out["slide_pred"].unsqueeze(0), out["attns"]



class MHA(ACMIL_MHA):
    """Drop-in for the reference's `MHA` (architecture/transformer.py:86-104; `--arch mha` of the generic trainer,
    Step3_WSI_classification.py): ONE single-query multi-head attention branch over the projected bag and one classifier -- ACMIL_MHA with
    one branch and without the bag head.  Same parameter names (`dimreduction`, `attention`, `q`, `classifier`), returns logits [1, C].
    Runs the folded single-query form of csrc/mha.hip through acmil_amd.autograd in every mode (the O(N) passes are HIP kernels with HIP
    backwards); Dropout(0.1) after out_proj applies in training mode as in the reference."""

    def __init__(self, conf, *, precision="f16x3"):
        nn.Module.__init__(self)
        self.dimreduction = DimReduction(conf.D_feat, conf.D_inner)
        self.attention = MutiHeadAttention(conf.D_inner, 8)
        self.q = nn.Parameter(torch.zeros((1, 1, conf.D_inner)))
        nn.init.normal_(self.q, std=1e-6)
        self.n_class = conf.n_class
        self.classifier = Classifier_1fc(conf.D_inner, conf.n_class, 0.0)
        self.n_token = 1
        self.precision = precision

    # the pieces ACMIL_MHA._forward_train walks over, under this module's own parameter names
    @property
    def sub_attention(self):
        return [self.attention]

    @property
    def bag_attention(self):
        return None

    def forward(self, input):
        if input.dim() != 3 or input.shape[0] != 1:
            raise RuntimeError("acmil_amd: MHA expects input [1, N, D_feat]")
        x = input[0]
        if not x.is_cuda:
            raise RuntimeError("acmil_amd: MHA runs on an MI355X only (no CPU fallback)")
        return self._forward_train(x.float().contiguous())[0]
