"""acmil_amd -- MI355X-native (gfx950) implementation of ACMIL's per-slide attention aggregation path.

Host code is Python on PyTorch-ROCm (device memory, streams, torch.distributed); the arithmetic is
hand-written HIP behind the C ABI of `libacmil_hip.so` (include/acmil_hip.h).  The sub-package
`acmil_amd.architecture` mirrors the reference's `architecture.{network,transformer,transMIL}` modules.
"""
from . import _lib  # noqa: F401

__all__ = ["_lib"]
