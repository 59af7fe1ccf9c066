"""CPU oracle for the ACMIL gated-attention aggregation path.  TEST INFRASTRUCTURE ONLY.

This file is a plain torch-CPU restatement (own code, functional form) of the reference's
per-slide gated-attention forward and of the ACMIL loss terms computed by its trainer.  Only
`tests/`, `__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import it; the
product path (`acmil_amd/`) never does.

Parity status: PINNED.  The reference has no tests or golden vectors of its own
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself:
`tests/golden/make_golden.py` imports `/root/reference` in the development container, runs
`architecture.transformer.{ACMIL_GA,ABMIL}` and the Step3 loss code on seeded inputs and
commits inputs + outputs as `.npz` fixtures; `tests/test_oracle_golden.py` checks this file
against every one of them (bit-exact on fp32 for the forward).

Weights are passed as a dict using the reference's `state_dict()` key names
(SURVEY.md section 8b), values torch tensors of one floating dtype.  Every function works in
whatever dtype it is handed: fp32 mirrors the reference, fp64 gives a ground truth used by the
tests to measure the error of BOTH the reference-precision oracle and the HIP kernels.
"""
from __future__ import annotations

import math
from typing import Dict, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


# --------------------------------------------------------------------------- leaf blocks
def dim_reduction(x: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """`DimReduction.forward` with numLayer_Res=0: relu(x @ W1^T), no bias.
    reference: architecture/network.py:49-57 (fc1 defined :40, bias=False)."""
    return F.relu(F.linear(x, sd["dimreduction.fc1.weight"]))


def attention_gated(h: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """`Attention_Gated.forward`: A = ((tanh(h Wv^T+bv) * sigmoid(h Wu^T+bu)) Ww^T + bw)^T -> [K,N].
    reference: architecture/transformer.py:259-267."""
    a_v = torch.tanh(F.linear(h, sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"]))
    a_u = torch.sigmoid(F.linear(h, sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"]))
    a = F.linear(a_v * a_u, sd["attention.attention_weights.weight"], sd["attention.attention_weights.bias"])
    return torch.transpose(a, 1, 0)


def classifier_1fc(v: Tensor, sd: Dict[str, Tensor], prefix: str) -> Tensor:
    """`Classifier_1fc.forward` with droprate=0 (the only value used on this path).
    reference: architecture/network.py:14-19."""
    return F.linear(v, sd[prefix + ".fc.weight"], sd[prefix + ".fc.bias"])


# --------------------------------------------------------------------------- STKIM mask-drop
def stkim_select(a: Tensor, n_masked_patch: int, mask_drop: float, uniforms: Tensor):
    """Top-k + random subset selection of STKIM, with the `torch.rand` draw injected.

    reference: architecture/transformer.py:311-317.  `uniforms` is what the reference draws
    with `torch.rand(*indices.shape)`, shape [K, k] with k = min(n_masked_patch, N).
    Returns (topk_indices [K,k] int64 sorted by descending score, masked_indices [K,m] int64)
    with m = int(k * mask_drop)."""
    k_branches, n = a.shape
    k = min(n_masked_patch, n)
    _, indices = torch.topk(a, k, dim=-1)
    m = int(k * mask_drop)
    rand_selected = torch.argsort(uniforms, dim=-1)[:, :m]
    masked_indices = indices[torch.arange(k_branches).unsqueeze(-1), rand_selected]
    return indices, masked_indices


def stkim_apply(a: Tensor, masked_indices: Tensor) -> Tensor:
    """scatter 0 into a ones-mask and masked_fill(-1e9).  reference: transformer.py:318-320."""
    mask = torch.ones_like(a)
    mask.scatter_(-1, masked_indices, 0)
    return a.masked_fill(mask == 0, -1e9)


# --------------------------------------------------------------------------- module forwards
def acmil_ga_forward(x: Tensor, sd: Dict[str, Tensor], n_token: int, n_masked_patch: int = 0,
                     mask_drop: float = 0.0, training: bool = False,
                     uniforms: Optional[Tensor] = None) -> Dict[str, Tensor]:
    """`ACMIL_GA.forward`.  reference: architecture/transformer.py:305-330.

    x: [1,N,D_feat] (B must be 1, the reference takes x[0]).  Returns a dict with the reference's
    three return values under 'sub_preds' [K,C], 'slide_pred' [1,C], 'A_out' [1,K,N] plus the
    intermediates the parity tests compare ('h', 'afeat', 'bag_feat', 'topk_idx', 'masked_idx')."""
    out: Dict[str, Tensor] = {}
    x = x[0]
    h = dim_reduction(x, sd)
    a = attention_gated(h, sd)
    if n_masked_patch > 0 and training:
        k = min(n_masked_patch, a.shape[1])
        if uniforms is None:
            uniforms = torch.rand(a.shape[0], k)
        idx, midx = stkim_select(a, n_masked_patch, mask_drop, uniforms)
        a = stkim_apply(a, midx)
        out["topk_idx"], out["masked_idx"] = idx, midx
    a_out = a
    p = F.softmax(a, dim=1)
    afeat = torch.mm(p, h)
    sub = torch.stack([classifier_1fc(afeat[i], sd, "classifier.%d" % i) for i in range(n_token)], dim=0)
    bag_a = F.softmax(a_out, dim=1).mean(0, keepdim=True)
    bag_feat = torch.mm(bag_a, h)
    out.update(sub_preds=sub, slide_pred=classifier_1fc(bag_feat, sd, "Slide_classifier"),
               A_out=a_out.unsqueeze(0), h=h, afeat=afeat, bag_feat=bag_feat)
    return out


def acmil_ga_forward_feature(x: Tensor, sd: Dict[str, Tensor], n_masked_patch: int = 0,
                             mask_drop: float = 0.0, use_attention_mask: bool = False,
                             uniforms: Optional[Tensor] = None) -> Tensor:
    """`ACMIL_GA.forward_feature` -> bag_feat [1,Di].  reference: transformer.py:332-352."""
    x = x[0]
    h = dim_reduction(x, sd)
    a = attention_gated(h, sd)
    if n_masked_patch > 0 and use_attention_mask:
        k = min(n_masked_patch, a.shape[1])
        if uniforms is None:
            uniforms = torch.rand(a.shape[0], k)
        _, midx = stkim_select(a, n_masked_patch, mask_drop, uniforms)
        a = stkim_apply(a, midx)
    bag_a = F.softmax(a, dim=1).mean(0, keepdim=True)
    return torch.mm(bag_a, h)


def abmil_forward(x: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """`ABMIL.forward` -> logits [1,C].  reference: architecture/transformer.py:277-286.
    state_dict differs from ACMIL_GA only in having one 'classifier.fc' head."""
    x = x[0]
    h = dim_reduction(x, sd)
    a = attention_gated(h, sd)
    p = F.softmax(a, dim=1)
    afeat = torch.mm(p, h)
    return classifier_1fc(afeat, sd, "classifier")


# --------------------------------------------------------------------------- trainer-side math
def acmil_losses(sub_preds: Tensor, slide_pred: Tensor, attn: Tensor, label: Tensor, n_token: int):
    """The three ACMIL loss terms of one training step.
    reference: Step3_WSI_classification_ACMIL.py:201-216.
    sub_preds [K,C], slide_pred [1,C], attn = A_out [1,K,N], label [1] int64.
    Returns (loss0, loss1, diff_loss)."""
    if n_token > 1:
        loss0 = F.cross_entropy(sub_preds, label.repeat_interleave(n_token))
    else:
        loss0 = torch.tensor(0.0)
    loss1 = F.cross_entropy(slide_pred, label)
    diff_loss = torch.zeros((), dtype=attn.dtype)
    p = torch.softmax(attn, dim=-1)
    for i in range(n_token):
        for j in range(i + 1, n_token):
            diff_loss = diff_loss + torch.cosine_similarity(p[:, i], p[:, j], dim=-1).mean() / (
                n_token * (n_token - 1) / 2)
    return loss0, loss1, diff_loss


def eval_div_loss(attn: Tensor) -> Tensor:
    """Entropy-style diagnostic of `evaluate()`.  reference: Step3_WSI_classification_ACMIL.py:259."""
    return torch.sum(F.softmax(attn, dim=-1) * F.log_softmax(attn, dim=-1)) / attn.shape[1]


def adjust_learning_rate(epoch: float, lr: float, min_lr: float, warmup_epoch: float, train_epoch: float) -> float:
    """Linear warm-up then half-cosine.  reference: utils/utils.py:250-262."""
    if epoch < warmup_epoch:
        return lr * epoch / warmup_epoch
    return min_lr + (lr - min_lr) * 0.5 * (1.0 + math.cos(math.pi * (epoch - warmup_epoch) / (train_epoch - warmup_epoch)))


# --------------------------------------------------------------------------- helpers for tests / bench
def default_state_dict(*args, **kwargs) -> Dict[str, Tensor]:
    """Synthetic GA / ABMIL parameters (generator shared with the benchmarks: acmil_amd/synthetic.py)."""
    from acmil_amd.synthetic import ga_state_dict
    return ga_state_dict(*args, **kwargs)


def synthetic_bag(*args, **kwargs) -> Tensor:
    from acmil_amd.synthetic import synthetic_bag as _bag
    return _bag(*args, **kwargs)
