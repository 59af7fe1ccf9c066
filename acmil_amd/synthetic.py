"""Deterministic synthetic weights and bags for benchmarks, smoke runs and tests: parameters drawn with
torch's default nn.Linear / nn.Conv2d init families under the reference's state_dict names, bags = randn(N, D) from
`manual_seed(1000 + slide_idx)`.  Pure data generation: no model arithmetic lives here."""
from __future__ import annotations

import math
from typing import Dict

import torch

Tensor = torch.Tensor
HEADS, RES_KERNEL = 8, 33      # transMIL.py:16, nystrom_attention.py:38


def ga_state_dict(d_feat: int, d_inner: int, n_class: int, n_token: int, d_attn: int = 128,
                       seed: int = 0, abmil: bool = False) -> Dict[str, Tensor]:
    """Weights with torch's default nn.Linear init (kaiming_uniform(a=sqrt(5)) == U(-1/sqrt(fan_in),
    +1/sqrt(fan_in)) for weight and bias), which is what the reference modules use (no custom
    init on this path).  Deterministic in `seed`; NOT draw-for-draw identical to
    constructing the reference module (golden fixtures carry the reference's own weights)."""
    g = torch.Generator().manual_seed(seed)

    def lin(out_f, in_f, bias=True):
        bound = 1.0 / math.sqrt(in_f)
        w = (torch.rand(out_f, in_f, generator=g) * 2 - 1) * bound
        b = (torch.rand(out_f, generator=g) * 2 - 1) * bound if bias else None
        return w, b

    sd: Dict[str, Tensor] = {}
    sd["dimreduction.fc1.weight"], _ = lin(d_inner, d_feat, bias=False)
    sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"] = lin(d_attn, d_inner)
    sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"] = lin(d_attn, d_inner)
    sd["attention.attention_weights.weight"], sd["attention.attention_weights.bias"] = lin(n_token, d_attn)
    if abmil:
        sd["classifier.fc.weight"], sd["classifier.fc.bias"] = lin(n_class, d_inner)
    else:
        for i in range(n_token):
            sd["classifier.%d.fc.weight" % i], sd["classifier.%d.fc.bias" % i] = lin(n_class, d_inner)
        sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"] = lin(n_class, d_inner)
    return sd


def synthetic_bag(n: int, d: int, slide_idx: int = 0, fp16_exact: bool = False) -> Tensor:
    """Synthetic bag of SURVEY.md 8(d): randn(N,D) fp32 from manual_seed(1000+slide_idx);
    the data-faithful variant rounds through fp16 (what Step2 stores, Step2_feature_extract.py:165)."""
    g = torch.Generator().manual_seed(1000 + slide_idx)
    x = torch.randn(n, d, generator=g)
    if fp16_exact:
        x = x.half().float()
    return x.unsqueeze(0)


def transmil_state_dict(d_feat: int, d_inner: int, n_class: int, seed: int = 0) -> Dict[str, Tensor]:
    """Random weights with the shapes / init families of the reference modules (nn.Linear / nn.Conv2d defaults,
    LayerNorm ones/zeros, cls_token ~ randn).  Deterministic in `seed`; not draw-identical to the reference ctor."""
    g = torch.Generator().manual_seed(seed)
    u = lambda shape, bound: (torch.rand(*shape, generator=g) * 2 - 1) * bound
    sd: Dict[str, Tensor] = {}
    sd["_fc1.0.weight"], sd["_fc1.0.bias"] = u((d_inner, d_feat), d_feat ** -0.5), u((d_inner,), d_feat ** -0.5)
    sd["cls_token"] = torch.randn(1, 1, d_inner, generator=g)
    for name, ksz in (("pos_layer.proj", 7), ("pos_layer.proj1", 5), ("pos_layer.proj2", 3)):
        bound = (ksz * ksz) ** -0.5
        sd[name + ".weight"], sd[name + ".bias"] = u((d_inner, 1, ksz, ksz), bound), u((d_inner,), bound)
    for layer in ("layer1", "layer2"):
        sd[layer + ".norm.weight"], sd[layer + ".norm.bias"] = torch.ones(d_inner), torch.zeros(d_inner)
        sd[layer + ".attn.to_qkv.weight"] = u((3 * d_inner, d_inner), d_inner ** -0.5)
        sd[layer + ".attn.to_out.0.weight"], sd[layer + ".attn.to_out.0.bias"] = u((d_inner, d_inner), d_inner ** -0.5), u((d_inner,), d_inner ** -0.5)
        sd[layer + ".attn.res_conv.weight"] = u((HEADS, 1, RES_KERNEL, 1), RES_KERNEL ** -0.5)
    sd["norm.weight"], sd["norm.bias"] = torch.ones(d_inner), torch.zeros(d_inner)
    sd["_fc2.weight"], sd["_fc2.bias"] = u((n_class, d_inner), d_inner ** -0.5), u((n_class,), d_inner ** -0.5)
    return sd
