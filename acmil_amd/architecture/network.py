"""Leaf blocks of the aggregators: same names / ctor signatures / state_dict keys as the reference's
`architecture/network.py` (Classifier_1fc :6-19, DimReduction :37-57).

Inside `ACMIL_GA` / `ABMIL` these modules are parameter containers only: their arithmetic is fused
into the HIP forward (acmil_amd/csrc/ga_forward_kernel.h).  Their own `forward` is kept for API
completeness (standalone use, not on the hot path) and issues plain library GEMMs through torch.
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


class DimReduction(nn.Module):
    """relu(x @ W1^T), bias-free.  numLayer_Res > 0 (unused by every shipped config) is not supported."""

    def __init__(self, n_channels, m_dim=512, numLayer_Res=0):
        super().__init__()
        if numLayer_Res != 0:
            raise NotImplementedError("acmil_amd: DimReduction residual blocks are outside the aggregation hot path")
        self.fc1 = nn.Linear(n_channels, m_dim, bias=False)
        self.numRes = numLayer_Res

    def forward(self, x):
        return F.relu(self.fc1(x))
