"""Leaf blocks of the aggregators: same names / ctor signatures / state_dict keys as the reference's
`architecture/network.py` (Classifier_1fc :6-19, residual_block :22-34, DimReduction :37-57).

Inside `ACMIL_GA` / `ABMIL` these modules are parameter containers: their arithmetic is fused into the HIP forward
(acmil_amd/csrc/ga_forward_kernel_v2.h).  Their own `forward` serves stand-alone use (DTFD's pipeline builds DimReduction by itself)
and the op-by-op path of the aggregators: on the GPU every Linear is a split-f16 MFMA GEMM of libacmil_hip.so with a HIP backward
(acmil_amd.autograd.linear); CPU tensors take torch's ops (parameter containers are usable in CPU-side tooling).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F


class Classifier_1fc(nn.Module):
    def __init__(self, n_channels, n_classes, droprate=0.0):
        super().__init__()
        self.fc = nn.Linear(n_channels, n_classes)
        self.droprate = droprate
        if self.droprate != 0.0:
            self.dropout = torch.nn.Dropout(p=self.droprate)

    def forward(self, x):
        if self.droprate != 0.0:
            x = self.dropout(x)
        return self.fc(x)


class residual_block(nn.Module):
    """x + relu(W2 relu(W1 x)), both Linear layers bias-free (network.py:22-34); state_dict keys block.0.weight / block.2.weight."""

    def __init__(self, nChn=512):
        super().__init__()
        self.block = nn.Sequential(nn.Linear(nChn, nChn, bias=False), nn.ReLU(inplace=True), nn.Linear(nChn, nChn, bias=False),
                                   nn.ReLU(inplace=True))

    def forward(self, x):
        if x.is_cuda:
            from .. import autograd as AG
            t = AG.linear(x, self.block[0].weight, None, relu=True)
            return x + AG.linear(t, self.block[2].weight, None, relu=True)
        return x + self.block(x)


class DimReduction(nn.Module):
    """relu(x @ W1^T), bias-free, followed by `numLayer_Res` residual blocks (network.py:37-57; every shipped ACMIL config uses 0,
    which is what the fused kernels cover -- with residual blocks the aggregators take their op-by-op path)."""

    def __init__(self, n_channels, m_dim=512, numLayer_Res=0):
        super().__init__()
        self.fc1 = nn.Linear(n_channels, m_dim, bias=False)
        self.numRes = numLayer_Res
        self.resBlocks = nn.Sequential(*[residual_block(m_dim) for _ in range(numLayer_Res)])

    def forward(self, x):
        if x.is_cuda:
            from .. import autograd as AG
            lead = x.shape[:-1]
            h = AG.linear(x.reshape(-1, x.shape[-1]).float(), self.fc1.weight, None, relu=True)
            if self.numRes > 0:
                h = self.resBlocks(h)
            return h.reshape(*lead, -1)
        h = F.relu(self.fc1(x))
        return self.resBlocks(h) if self.numRes > 0 else h
