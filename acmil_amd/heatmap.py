"""Attention heat-map scores of a slide -- the model-side half of the reference's heat-map export
(Step4_visualize_heatmap_camelyon.py:110-121): forward the bag, softmax the raw scores over the patches, average the
branches, scale by N * zoom_factor (* 100 at the drawing call).  The WSI rendering (`WholeSlideImage.visHeatmap`) is outside
the aggregation path; this returns the per-patch scores it is fed with, computed by acmil_attn_heatmap on the device."""
from __future__ import annotations

import torch

from . import ops


@torch.no_grad()
def heatmap_scores(net, feat: torch.Tensor, zoom_factor: float = 1.0) -> torch.Tensor:
    """net: ACMIL_GA / ACMIL_MHA-style module in eval mode returning (sub_preds, slide_pred, attn [1,K,N]);
    feat [1,N,D_feat] on the GPU.  Returns `probs * 100` of Step4:117-119 as a [N] device tensor."""
    _, _, attn = net(feat)
    return ops.attn_heatmap(attn, zoom_factor) * 100.0


@torch.no_grad()
def branch_heatmap_scores(attn: torch.Tensor, branch: int, zoom_factor: float = 1.0) -> torch.Tensor:
    """Per-branch variant (the commented block Step4:123-127): softmax(attn, -1)[0, branch] * N * zoom_factor * 100."""
    return ops.attn_heatmap(attn[:, branch:branch + 1], zoom_factor) * 100.0
