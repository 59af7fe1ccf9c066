"""torch.autograd Functions over the HIP ops: the building blocks of the TRAINING paths of TransMIL, ACMIL_MHA, the DTFD
attention blocks and IBMIL (ACMIL_GA / ABMIL have their own fused backward: csrc/ga_backward.hip).

Each Function's forward and backward are C-ABI kernels (csrc/gemm_f32.hip, csrc/transmil_train.hip, csrc/attn_generic.hip);
autograd only sequences them and handles views / concatenation.  No torch math kernel runs over an O(N) tensor except
copies (cat / pad / slicing / `+`) and the dropout mask.  CUDA fp32 tensors only (no CPU fallback).

  linear(x, W, b, relu)          nn.Linear (+ReLU)                    -> acmil_gemm_*            (transMIL.py:51,62; nystrom_attention.py:55,59)
  matmul(a, b, trans_a, trans_b) batched aten::matmul                 -> acmil_gemm_*            (nystrom_attention.py:113-133)
  layer_norm(x, g, b, eps)       nn.LayerNorm                         -> acmil_layernorm_fwd/_bwd
  softmax_rows(s)                softmax(dim=-1)                      -> acmil_softmax_rows/_bwd
  seq_conv(v, w)                 Conv2d(8, 8, (33,1), groups=8)       -> acmil_seqconv/_bwd_w
  dwconv7(x, weff, beff, side)   folded PPEG depth-wise stencil       -> acmil_dwconv7/_bwd_w
  landmark_mean(src, l)          reduce(..., 'sum') / l               -> acmil_landmark_mean/_bwd
  gated_scores(h, ...)           Attention_Gated / Attn_Net_Gated     -> linear x3 + acmil_gate_fwd/_bwd
  attn_pool(h, A)                softmax over N, then P @ h           -> softmax_rows + matmul
"""
from __future__ import annotations

from typing import Optional

import torch

from . import _lib, ops

_GRAD_PREC = {"fp32": "fp32", "f16x3": "bf16x3", "bf16x3": "bf16x3"}   # gradient operands reach 1e-8: bf16 halves keep the exponent


def _stream():
    return torch.cuda.current_stream().cuda_stream


def _c(t: torch.Tensor) -> torch.Tensor:
    if not t.is_cuda or t.dtype != torch.float32:
        raise RuntimeError("acmil_amd.autograd: CUDA fp32 tensors only (no CPU fallback)")
    return t


def _rows(t: torch.Tensor) -> torch.Tensor:
    """tensor usable as a GEMM operand: inner stride 1 (2-D / 3-D); anything else is made contiguous"""
    return t if (t.stride(-1) == 1 or t.shape[-1] == 1) else t.contiguous()


def _colsum(x2: torch.Tensor) -> torch.Tensor:
    lib = _lib.load()
    rows, cols = x2.shape
    out = torch.empty(cols, dtype=torch.float32, device=x2.device)
    ws = torch.empty(lib.acmil_colsum_workspace_bytes(rows, cols), dtype=torch.uint8, device=x2.device)
    _lib.check(lib.acmil_colsum(x2.data_ptr(), rows, cols, out.data_ptr(), ws.data_ptr(), _stream()), "acmil_colsum")
    return out


class _Matmul(torch.autograd.Function):
    """out = alpha * op(a) @ op(b) (+ bias) (+ relu); 2-D or batched 3-D; b may be broadcast over the batch (b.dim() == 2)."""

    @staticmethod
    def forward(ctx, a, b, bias, trans_a, trans_b, alpha, relu, precision):
        a, b = _rows(_c(a.detach())), _rows(_c(b.detach()))
        out = ops.gemm(a, b if (b.dim() == a.dim()) else b.unsqueeze(0), trans_a=trans_a, trans_b=trans_b, alpha=alpha,
                       bias=None if bias is None else bias.detach(), act=1 if relu else 0, precision=precision)
        ctx.save_for_backward(a, b, out if relu else None)
        ctx.cfg = (trans_a, trans_b, alpha, relu, precision, bias is not None)
        return out

    @staticmethod
    def backward(ctx, dout):
        a, b, out = ctx.saved_tensors
        ta, tb, alpha, relu, precision, has_bias = ctx.cfg
        gp = _GRAD_PREC[precision]
        dz = _c(dout).contiguous()
        if relu:
            lib = _lib.load()
            dm = torch.empty_like(dz)
            _lib.check(lib.acmil_relu_bwd(dz.data_ptr(), out.data_ptr(), dm.data_ptr(), dz.numel(), _stream()), "acmil_relu_bwd")
            dz = dm
        bb = b if b.dim() == a.dim() else b.unsqueeze(0)
        da = db = dbias = None
        if ctx.needs_input_grad[0]:
            da = ops.gemm(dz, bb, trans_b=not tb, alpha=alpha, precision=gp) if not ta else \
                ops.gemm(bb, dz, trans_a=tb, trans_b=True, alpha=alpha, precision=gp)
        if ctx.needs_input_grad[1]:
            db = ops.gemm(a, dz, trans_a=not ta, alpha=alpha, precision=gp) if not tb else \
                ops.gemm(dz, a, trans_a=True, trans_b=ta, alpha=alpha, precision=gp)
            if b.dim() != a.dim():                      # b was broadcast over the batch: sum the per-batch gradients
                db = db.sum(0) if db.shape[0] > 1 else db[0]
        if has_bias and ctx.needs_input_grad[2]:
            dbias = _colsum(dz.reshape(-1, dz.shape[-1]))
        return da, db, dbias, None, None, None, None, None


def matmul(a, b, trans_a=False, trans_b=False, alpha=1.0, precision="fp32"):
    return _Matmul.apply(a, b, None, trans_a, trans_b, float(alpha), False, precision)


def linear(x, weight, bias=None, relu=False, precision="f16x3"):
    """x [rows, in] @ weight[out, in]^T + bias"""
    return _Matmul.apply(x, weight, bias, False, True, 1.0, relu, precision)


class _LayerNorm(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, gamma, beta, eps):
        lib = _lib.load()
        x2 = _c(x.detach()).contiguous().reshape(-1, x.shape[-1])
        rows, dim = x2.shape
        y = torch.empty_like(x2)
        stats = torch.empty(rows, 2, dtype=torch.float32, device=x2.device)
        g, b = gamma.detach().contiguous(), beta.detach().contiguous()
        _lib.check(lib.acmil_layernorm_fwd(x2.data_ptr(), rows, dim, g.data_ptr(), b.data_ptr(), float(eps), y.data_ptr(),
                                           stats.data_ptr(), _stream()), "acmil_layernorm_fwd")
        ctx.save_for_backward(x2, stats, g)
        return y.reshape(x.shape)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x2, stats, g = ctx.saved_tensors
        rows, dim = x2.shape
        dy2 = _c(dy).contiguous().reshape(rows, dim)
        dx = torch.empty_like(x2)
        dg, db = torch.empty(dim, dtype=torch.float32, device=x2.device), torch.empty(dim, dtype=torch.float32, device=x2.device)
        ws = torch.empty(lib.acmil_layernorm_bwd_workspace_bytes(rows, dim), dtype=torch.uint8, device=x2.device)
        _lib.check(lib.acmil_layernorm_bwd(x2.data_ptr(), dy2.data_ptr(), stats.data_ptr(), g.data_ptr(), rows, dim, dx.data_ptr(),
                                           dg.data_ptr(), db.data_ptr(), ws.data_ptr(), _stream()), "acmil_layernorm_bwd")
        return dx.reshape(dy.shape), dg, db, None


def layer_norm(x, gamma, beta, eps=1e-5):
    return _LayerNorm.apply(x, gamma, beta, eps)


class _SoftmaxRows(torch.autograd.Function):
    @staticmethod
    def forward(ctx, s):
        s2 = _c(s.detach()).contiguous().reshape(-1, s.shape[-1])
        p = ops.softmax_rows(s2)
        ctx.save_for_backward(p)
        return p.reshape(s.shape)

    @staticmethod
    def backward(ctx, dp):
        lib = _lib.load()
        (p,) = ctx.saved_tensors
        dp2 = _c(dp).contiguous().reshape(p.shape)
        ds = torch.empty_like(p)
        _lib.check(lib.acmil_softmax_rows_bwd(p.data_ptr(), dp2.data_ptr(), ds.data_ptr(), p.shape[0], p.shape[1], _stream()),
                   "acmil_softmax_rows_bwd")
        return ds.reshape(dp.shape)


def softmax_rows(s):
    return _SoftmaxRows.apply(s)


def _seqconv_raw(v: torch.Tensor, w: torch.Tensor) -> torch.Tensor:
    """v [n, Di] (row stride free, inner stride 1), w [8, 33] -> [n, Di]"""
    lib = _lib.load()
    n, di = v.shape
    out = torch.empty(n, di, dtype=torch.float32, device=v.device)
    _lib.check(lib.acmil_seqconv(v.data_ptr(), v.stride(0), n, di, w.data_ptr(), out.data_ptr(), _stream()), "acmil_seqconv")
    return out


class _SeqConv(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, w):
        v = _c(v.detach())
        if v.stride(1) != 1 or v.stride(0) % 4 != 0 or v.data_ptr() % 16 != 0:
            v = v.contiguous()
        w2 = w.detach().reshape(8, 33).contiguous()
        ctx.save_for_backward(v, w2)
        ctx.wshape = w.shape
        return _seqconv_raw(v, w2)

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        v, w2 = ctx.saved_tensors
        dout = _c(dout).contiguous()
        n, di = dout.shape
        dv = dw = None
        if ctx.needs_input_grad[0]:
            dv = _seqconv_raw(dout, torch.flip(w2, dims=(1,)).contiguous())      # correlation with the flipped kernel
        if ctx.needs_input_grad[1]:
            dw = torch.empty(8, 33, dtype=torch.float32, device=dout.device)
            ws = torch.empty(lib.acmil_seqconv_bwd_w_workspace_bytes(n, di), dtype=torch.uint8, device=dout.device)
            _lib.check(lib.acmil_seqconv_bwd_w(dout.data_ptr(), v.data_ptr(), v.stride(0), n, di, dw.data_ptr(), ws.data_ptr(), _stream()),
                       "acmil_seqconv_bwd_w")
            dw = dw.reshape(ctx.wshape)
        return dv, dw


def seq_conv(v, w):
    return _SeqConv.apply(v, w)


def _dwconv7_raw(x, weff, beff, side):
    lib = _lib.load()
    y = torch.empty_like(x)
    _lib.check(lib.acmil_dwconv7(x.data_ptr(), side, x.shape[1], weff.data_ptr(), None if beff is None else beff.data_ptr(), y.data_ptr(),
                                 _stream()), "acmil_dwconv7")
    return y


class _DwConv7(torch.autograd.Function):
    """x [side*side, C] channels-last token grid, weff [49, C] (tap-major), beff [C]"""

    @staticmethod
    def forward(ctx, x, weff, beff, side):
        x, weff, beff = _c(x.detach()).contiguous(), weff.detach().contiguous(), beff.detach().contiguous()
        ctx.save_for_backward(x, weff)
        ctx.side = side
        return _dwconv7_raw(x, weff, beff, side)

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        x, weff = ctx.saved_tensors
        side, c = ctx.side, x.shape[1]
        dy = _c(dy).contiguous()
        dx = dw = db = None
        if ctx.needs_input_grad[0]:
            dx = _dwconv7_raw(dy, torch.flip(weff, dims=(0,)).contiguous(), None, side)   # tap (ky,kx) -> (6-ky,6-kx) = reversed tap index
        if ctx.needs_input_grad[1] or ctx.needs_input_grad[2]:
            dw = torch.empty(49, c, dtype=torch.float32, device=x.device)
            db = torch.empty(c, dtype=torch.float32, device=x.device)
            ws = torch.empty(lib.acmil_dwconv7_bwd_w_workspace_bytes(side, c), dtype=torch.uint8, device=x.device)
            _lib.check(lib.acmil_dwconv7_bwd_w(dy.data_ptr(), x.data_ptr(), side, c, dw.data_ptr(), db.data_ptr(), ws.data_ptr(), _stream()),
                       "acmil_dwconv7_bwd_w")
        return dx, dw, db, None


def dwconv7(x, weff, beff, side):
    return _DwConv7.apply(x, weff, beff, side)


class _LandmarkMean(torch.autograd.Function):
    """src [n, Di] (row stride free) -> [8, n / l, Di / 8]"""

    @staticmethod
    def forward(ctx, src, l):
        lib = _lib.load()
        src = _c(src.detach())
        if src.stride(1) != 1 or src.stride(0) % 4 != 0 or src.data_ptr() % 16 != 0:
            src = src.contiguous()
        n, di = src.shape
        out = torch.empty(8, n // l, di // 8, dtype=torch.float32, device=src.device)
        _lib.check(lib.acmil_landmark_mean(src.data_ptr(), src.stride(0), n, l, di, out.data_ptr(), _stream()), "acmil_landmark_mean")
        ctx.cfg = (n, l, di)
        return out

    @staticmethod
    def backward(ctx, dout):
        lib = _lib.load()
        n, l, di = ctx.cfg
        dout = _c(dout).contiguous()
        dsrc = torch.empty(n, di, dtype=torch.float32, device=dout.device)
        _lib.check(lib.acmil_landmark_mean_bwd(dout.data_ptr(), n, l, di, dsrc.data_ptr(), _stream()), "acmil_landmark_mean_bwd")
        return dsrc, None


def landmark_mean(src, l):
    return _LandmarkMean.apply(src, l)


class _Gate(torch.autograd.Function):
    """G [N, 2 Da] -> tanh(G[:, :Da]) * sigmoid(G[:, Da:])"""

    @staticmethod
    def forward(ctx, G):
        lib = _lib.load()
        G = _c(G.detach()).contiguous()
        n, da = G.shape[0], G.shape[1] // 2
        y = torch.empty(n, da, dtype=torch.float32, device=G.device)
        _lib.check(lib.acmil_gate_fwd(G.data_ptr(), y.data_ptr(), n, da, _stream()), "acmil_gate_fwd")
        ctx.save_for_backward(G)
        return y

    @staticmethod
    def backward(ctx, dy):
        lib = _lib.load()
        (G,) = ctx.saved_tensors
        dy = _c(dy).contiguous()
        dG = torch.empty_like(G)
        _lib.check(lib.acmil_gate_bwd(G.data_ptr(), dy.data_ptr(), dG.data_ptr(), G.shape[0], G.shape[1] // 2, _stream()), "acmil_gate_bwd")
        return dG


def gated_scores(h, Wv, bv, Wu, bu, Ww, bw, precision="f16x3"):
    """Differentiable gated-attention scores A [K, N] of a projected bag h [N, L] (Attention_Gated / Attn_Net_Gated):
    two Linear products into the halves of G, the gate kernel, one Linear, a transpose."""
    G = torch.cat([linear(h, Wv, bv, precision=precision), linear(h, Wu, bu, precision=precision)], dim=1)
    return linear(_Gate.apply(G), Ww, bw, precision="fp32").t()


def attn_pool(h, A):
    """softmax over N of A [K, N], then P @ h -> [K, Di] (differentiable)"""
    return matmul(softmax_rows(A.contiguous()), h)


class _MaskFill(torch.autograd.Function):
    """STKIM mask as a differentiable op: out = S with -1e9 at (row, midx[row, j]); masked entries get zero gradient
    (`attn.masked_fill(random_mask == 0, -1e9)`, architecture/transformer.py:168-171 / :318-320).  The index set is tiny
    ([rows, m]); the only O(N) work is one copy of S."""

    @staticmethod
    def forward(ctx, S, midx):
        ctx.save_for_backward(midx)
        out = S.clone()
        out.scatter_(1, midx, -1e9)
        return out

    @staticmethod
    def backward(ctx, g):
        (midx,) = ctx.saved_tensors
        g = g.clone()
        g.scatter_(1, midx, 0.0)
        return g, None


def mask_fill(S: torch.Tensor, midx: torch.Tensor) -> torch.Tensor:
    return _MaskFill.apply(S, midx)

