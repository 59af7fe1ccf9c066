"""CPU oracle for ACMIL_MHA (the `--arch mha` twin).  TEST INFRASTRUCTURE ONLY.

Plain torch-CPU restatement (own code, functional, state_dict-keyed) of the EVAL forward of
  ACMIL_MHA.forward                  architecture/transformer.py:68-83
  MutiHeadAttention.forward          architecture/transformer.py:142-185   (eval: no top-k masking, Dropout = identity)
  MutiHeadAttention_modify.forward   architecture/transformer.py:221-236
  DimReduction / Classifier_1fc      architecture/network.py:49-57, :14-19
It follows the reference's association (full k / v projections per branch), i.e. it does NOT use the single-query folding
of the HIP path, so the parity tests also check that algebra.  Pinned: tests/golden/make_golden_mha.py runs the real
reference module here and commits inputs / outputs; tests/test_oracle_mha.py checks this file against them.
Only `tests/`, `__graft_entry__.smoke()` and `bench.py`'s cpu_baseline leg may import it.
"""
from __future__ import annotations

import math
from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
HEADS = 8   # transformer.py:55,57


def _heads(t: Tensor) -> Tensor:          # [n, Di] -> [HEADS, n, c]        (transformer.py:132-135)
    n, di = t.shape
    return t.reshape(n, HEADS, di // HEADS).transpose(0, 1)


def _post(out1: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """recombine heads, out_proj, (dropout = identity), LayerNorm(eps 1e-6)    (transformer.py:178-183)"""
    di = out1.shape[0] * out1.shape[2]
    o = out1.transpose(0, 1).reshape(out1.shape[1], di)
    o = F.linear(o, sd[prefix + ".out_proj.weight"], sd[prefix + ".out_proj.bias"])
    return F.layer_norm(o, (di,), sd[prefix + ".layer_norm.weight"], sd[prefix + ".layer_norm.bias"], 1e-6)


def sub_attention(q: Tensor, h: Tensor, sd: Dict[str, Tensor], prefix: str):
    """q [1, Di], h [N, Di] -> (feat [1, Di], attn [HEADS, 1, N])"""
    qq = _heads(F.linear(q, sd[prefix + ".q_proj.weight"], sd[prefix + ".q_proj.bias"]))
    kk = _heads(F.linear(h, sd[prefix + ".k_proj.weight"], sd[prefix + ".k_proj.bias"]))
    vv = _heads(F.linear(h, sd[prefix + ".v_proj.weight"], sd[prefix + ".v_proj.bias"]))
    attn = qq @ kk.transpose(1, 2) / math.sqrt(qq.shape[-1])          # [HEADS, 1, N]
    out1 = torch.softmax(attn, dim=-1) @ vv                            # [HEADS, 1, c]
    return _post(out1, sd, prefix), attn


def acmil_mha_forward(x: Tensor, sd: Dict[str, Tensor], n_token: int) -> Dict[str, Tensor]:
    """x [1, N, D_feat] -> {'sub_preds' [K,C], 'slide_pred' [1,C], 'attns' [HEADS,K,N]}   (transformer.py:68-83)"""
    h = F.relu(F.linear(x[0], sd["dimreduction.fc1.weight"]))
    outs, attns = [], []
    for i in range(n_token):
        feat, attn = sub_attention(sd["q"][0, i].unsqueeze(0), h, sd, "sub_attention.%d" % i)
        outs.append(F.linear(feat, sd["classifier.%d.fc.weight" % i], sd["classifier.%d.fc.bias" % i]))
        attns.append(attn)
    attns = torch.cat(attns, 1)                                          # [HEADS, K, N]
    bag_attn = attns.softmax(dim=-1).mean(1, keepdim=True)               # [HEADS, 1, N]
    vv = _heads(F.linear(h, sd["bag_attention.v_proj.weight"], sd["bag_attention.v_proj.bias"]))
    feat_bag = _post(bag_attn @ vv, sd, "bag_attention")
    slide = F.linear(feat_bag, sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"])
    return {"sub_preds": torch.cat(outs, 0), "slide_pred": slide, "attns": attns}


def default_state_dict(d_feat: int, d_inner: int, n_class: int, n_token: int, seed: int = 0, q_std: float = 0.5) -> Dict[str, Tensor]:
    """Random parameters with the reference's names / shapes (nn.Linear default init families).  `q` is drawn with a
    visible std (the reference initialises it to N(0, 1e-6), transformer.py:59, which makes the query term invisible
    next to the q_proj bias; a trained q is not small)."""
    g = torch.Generator().manual_seed(seed)
    u = lambda shape, bound: (torch.rand(*shape, generator=g) * 2 - 1) * bound
    sd: Dict[str, Tensor] = {"q": torch.randn(1, n_token, d_inner, generator=g) * q_std,
                             "dimreduction.fc1.weight": u((d_inner, d_feat), d_feat ** -0.5)}
    b = d_inner ** -0.5
    def attn(prefix, full):
        for name in (("q_proj", "k_proj", "v_proj", "out_proj") if full else ("v_proj", "out_proj")):
            sd["%s.%s.weight" % (prefix, name)] = u((d_inner, d_inner), b)
            sd["%s.%s.bias" % (prefix, name)] = u((d_inner,), b)
        sd[prefix + ".layer_norm.weight"] = 1.0 + 0.1 * torch.randn(d_inner, generator=g)
        sd[prefix + ".layer_norm.bias"] = 0.1 * torch.randn(d_inner, generator=g)
    for i in range(n_token):
        attn("sub_attention.%d" % i, True)
    attn("bag_attention", False)
    for i in range(n_token):
        sd["classifier.%d.fc.weight" % i], sd["classifier.%d.fc.bias" % i] = u((n_class, d_inner), b), u((n_class,), b)
    sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"] = u((n_class, d_inner), b), u((n_class,), b)
    return sd
