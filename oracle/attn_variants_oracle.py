"""CPU oracle for the other gated-attention consumers.  TEST INFRASTRUCTURE ONLY.

Plain torch-CPU restatements (own code, functional, state_dict-keyed) of the eval forwards of
  Attention_Gated / Attention_with_Classifier   architecture/Attention.py:29-70
  IBMIL (confounder_path None)                  architecture/ibmil.py:69-113
  CLAM_SB (gate=True)                           architecture/clam.py:159-197
Pinned: tests/golden/make_golden_variants.py runs the real reference modules here and commits inputs / outputs;
tests/test_oracle_variants.py checks this file against them.  Only `tests/` may import it.
"""
from __future__ import annotations

from typing import Dict

import torch
import torch.nn.functional as F

Tensor = torch.Tensor


def gated_scores(x: Tensor, wv, bv, wu, bu, ww, bw) -> Tensor:
    """x [N,L] -> A [K,N] raw   (Attention.py:47-52)"""
    return F.linear(torch.tanh(F.linear(x, wv, bv)) * torch.sigmoid(F.linear(x, wu, bu)), ww, bw).transpose(0, 1)


def attention_gated(x: Tensor, sd: Dict[str, Tensor], prefix: str = "", is_norm: bool = True) -> Tensor:
    a = gated_scores(x, sd[prefix + "attention_V.0.weight"], sd[prefix + "attention_V.0.bias"], sd[prefix + "attention_U.0.weight"],
                     sd[prefix + "attention_U.0.bias"], sd[prefix + "attention_weights.weight"], sd[prefix + "attention_weights.bias"])
    return F.softmax(a, dim=1) if is_norm else a


def attention_with_classifier(x: Tensor, sd: Dict[str, Tensor]) -> Tensor:
    """x [N,L] -> pred [K,num_cls]   (Attention.py:66-70)"""
    afeat = attention_gated(x, sd, "attention.") @ x
    return F.linear(afeat, sd["classifier.fc.weight"], sd["classifier.fc.bias"])


def ibmil_forward(x: Tensor, sd: Dict[str, Tensor]):
    """x [1,N,D] -> (Y_prob [1,C], M [1,Di], A [1,N])   (ibmil.py:69-74,108-113, no confounder)"""
    h = F.relu(F.linear(x[0], sd["dimreduction.fc1.weight"]))
    a = attention_gated(h, sd, "attention.", is_norm=True)
    m = a @ h
    return F.linear(m, sd["classifier.fc.weight"], sd["classifier.fc.bias"]), m, a


def clam_sb_forward(x: Tensor, sd: Dict[str, Tensor], net_idx: int = 3):
    """x [1,N,L] -> (logits [1,C], A_raw [1,N], M [1,size1])   (clam.py:159-165,190-191; eval: dropout = identity)"""
    h = F.relu(F.linear(x[0], sd["attention_net.0.weight"], sd["attention_net.0.bias"]))
    p = "attention_net.%d." % net_idx
    a = gated_scores(h, sd[p + "attention_a.0.weight"], sd[p + "attention_a.0.bias"], sd[p + "attention_b.0.weight"],
                     sd[p + "attention_b.0.bias"], sd[p + "attention_c.weight"], sd[p + "attention_c.bias"])
    m = F.softmax(a, dim=-1) @ h
    return F.linear(m, sd["classifiers.weight"], sd["classifiers.bias"]), a, m
