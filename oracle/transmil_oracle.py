"""CPU oracle for the TransMIL / Nystrom-attention variant of the aggregation path.  TEST INFRASTRUCTURE ONLY.

Plain torch-CPU restatement (own code, functional, einops-free) of
  TransMIL.forward                architecture/transMIL.py:60-91
  TransLayer.forward              architecture/transMIL.py:25-28
  PPEG.forward                    architecture/transMIL.py:38-45
  NystromAttention.forward        architecture/nystrom_attention.py:67-149   (mask=None, return_attn=False)
  moore_penrose_iter_pinv         architecture/nystrom_attention.py:12-27
in eval mode (the Dropout(0.1) of `to_out` is the only train-time difference).  Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it.

Parity status: pinned against the reference AS VENDORED -- `tests/golden/make_golden_transmil.py` runs
`architecture.transMIL.TransMIL` from /root/reference with two harness-side shims (the un-vendored pip package
`nystrom_attention==0.0.12`, requirements.txt:42, is aliased to the repo's own fork
`architecture/nystrom_attention.py`; `Tensor.cuda` is a no-op on CPU) and commits inputs / outputs.  Parity at
the boundary of the pip wheel itself is UNPINNED (the wheel is not available offline); the fork implements the
same published algorithm (lucidrains/nystrom-attention) for the `return_attn=False` path used here.

Weights: dict with the reference's state_dict keys (`_fc1.0.weight`, `cls_token`, `layer1.norm.weight`,
`layer1.attn.to_qkv.weight`, `layer1.attn.to_out.0.weight`, `layer1.attn.res_conv.weight`, `pos_layer.proj*.weight`,
`norm.weight`, `_fc2.weight`, ...).
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
HEADS = 8          # transMIL.py:16
PINV_ITERS = 6     # transMIL.py:18
RES_KERNEL = 33    # nystrom_attention.py:38


def moore_penrose_iter_pinv(x: Tensor, iters: int = PINV_ITERS) -> Tensor:
    """nystrom_attention.py:12-27.  Note the init divides by the GLOBAL max over batch and heads."""
    abs_x = torch.abs(x)
    col = abs_x.sum(dim=-1)
    row = abs_x.sum(dim=-2)
    z = x.transpose(-1, -2) / (torch.max(col) * torch.max(row))
    eye = torch.eye(x.shape[-1], dtype=x.dtype).unsqueeze(0)
    for _ in range(iters):
        xz = x @ z
        z = 0.25 * z @ (13 * eye - (xz @ (15 * eye - (xz @ (7 * eye - xz)))))
    return z


def nystrom_attention(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """nystrom_attention.py:67-149 with heads=8, dim_head=dim/8, num_landmarks=dim/2, residual conv 33 (transMIL.py:13-23)."""
    b, n, dim = x.shape
    h, m = HEADS, dim // 2
    d = dim // h
    scale = d ** -0.5
    remainder = n % m
    if remainder > 0:
        x = F.pad(x, (0, 0, m - remainder, 0), value=0)          # FRONT zero padding (:72-75)
    npad = x.shape[1]
    qkv = F.linear(x, sd[prefix + ".to_qkv.weight"])
    q, k, v = qkv.chunk(3, dim=-1)
    split = lambda t: t.reshape(b, npad, h, d).permute(0, 2, 1, 3)   # b h n d
    q, k, v = split(q), split(k), split(v)
    q = q * scale
    l = math.ceil(n / m)
    q_l = q.reshape(b, h, m, l, d).sum(dim=3) / l                 # landmark means over l consecutive tokens (:95-111)
    k_l = k.reshape(b, h, m, l, d).sum(dim=3) / l
    sim1 = q @ k_l.transpose(-1, -2)
    sim2 = q_l @ k_l.transpose(-1, -2)
    sim3 = q_l @ k.transpose(-1, -2)
    attn1, attn2, attn3 = sim1.softmax(dim=-1), sim2.softmax(dim=-1), sim3.softmax(dim=-1)
    attn2 = moore_penrose_iter_pinv(attn2, PINV_ITERS)
    out = (attn1 @ attn2) @ (attn3 @ v)
    out = out + F.conv2d(v, sd[prefix + ".res_conv.weight"], padding=(RES_KERNEL // 2, 0), groups=h)   # (:135-136)
    out = out.permute(0, 2, 1, 3).reshape(b, npad, h * d)
    out = F.linear(out, sd[prefix + ".to_out.0.weight"], sd[prefix + ".to_out.0.bias"])
    return out[:, -n:]


def trans_layer(x: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """transMIL.py:25-28: x + attn(LayerNorm(x))."""
    dim = x.shape[-1]
    xn = F.layer_norm(x, (dim,), sd[prefix + ".norm.weight"], sd[prefix + ".norm.bias"], 1e-5)
    return x + nystrom_attention(xn, sd, prefix + ".attn")


def ppeg(x: Tensor, hh: int, ww: int, sd: Dict[str, Tensor]) -> Tensor:
    """transMIL.py:38-45: cls passthrough, depth-wise 7x7 + 5x5 + 3x3 convs (+identity) on the token grid."""
    b, _, c = x.shape
    cls_token, feat = x[:, 0], x[:, 1:]
    img = feat.transpose(1, 2).reshape(b, c, hh, ww)
    y = (F.conv2d(img, sd["pos_layer.proj.weight"], sd["pos_layer.proj.bias"], 1, 3, groups=c) + img +
         F.conv2d(img, sd["pos_layer.proj1.weight"], sd["pos_layer.proj1.bias"], 1, 2, groups=c) +
         F.conv2d(img, sd["pos_layer.proj2.weight"], sd["pos_layer.proj2.bias"], 1, 1, groups=c))
    y = y.flatten(2).transpose(1, 2)
    return torch.cat((cls_token.unsqueeze(1), y), dim=1)


def transmil_forward(x: Tensor, sd: Dict[str, Tensor]) -> Dict[str, Tensor]:
    """transMIL.py:60-91.  x [B,N,D_feat] -> dict(logits [B,C], plus intermediates h1 / hp / h2 for the parity tests)."""
    h = F.relu(F.linear(x, sd["_fc1.0.weight"], sd["_fc1.0.bias"]))
    n = h.shape[1]
    side = int(math.ceil(math.sqrt(n)))
    add = side * side - n
    h = torch.cat([h, h[:, :add, :]], dim=1)                      # pad by REPEATING the first tokens (:64-67)
    b = h.shape[0]
    h = torch.cat((sd["cls_token"].expand(b, -1, -1), h), dim=1)
    h1 = trans_layer(h, sd, "layer1")
    hp = ppeg(h1, side, side, sd)
    h2 = trans_layer(hp, sd, "layer2")
    dim = h2.shape[-1]
    cls = F.layer_norm(h2, (dim,), sd["norm.weight"], sd["norm.bias"], 1e-5)[:, 0]
    logits = F.linear(cls, sd["_fc2.weight"], sd["_fc2.bias"])
    return {"logits": logits, "h1": h1, "hp": hp, "h2": h2}


def default_state_dict(*args, **kwargs) -> Dict[str, Tensor]:
    """Synthetic TransMIL parameters (generator shared with the benchmarks: acmil_amd/synthetic.py)."""
    from acmil_amd.synthetic import transmil_state_dict
    return transmil_state_dict(*args, **kwargs)
