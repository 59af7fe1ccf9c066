import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ["ACMIL_LIN64"] = "1"
import torch
from acmil_amd import ops
torch.manual_seed(0)
M, K, N = 512, 128, 128
x = torch.randn(M, K, device="cuda")
for (c0, k0) in [(0, 0), (5, 3), (37, 9), (100, 70), (127, 127), (64, 16)]:
    w = torch.zeros(N, K, device="cuda"); w[c0, k0] = 1.0
    y = ops.linear_f16x3(x, ops.linear_pack(w), N)
    col = y[:, c0]
    # which k does each row's value match?
    d = (col[:, None] - x).abs()            # [M, K]
    kbest = d.argmin(1)
    ok = (kbest == k0) & (d.min(1).values < 1e-5)
    other = (y.abs().sum(1) - col.abs())
    print("c0=%d k0=%d: rows ok %d / %d; wrong rows sample %s kbest %s; energy in other columns %.3g" % (
        c0, k0, int(ok.sum()), M, (~ok).nonzero()[:8, 0].tolist(), kbest[~ok][:8].tolist(), other.abs().max().item()))
# full random check per row block
w = torch.randn(N, K, device="cuda") * 0.1
y = ops.linear_f16x3(x, ops.linear_pack(w), N)
ref = x.double() @ w.double().T
err = (y.double() - ref).abs()
print("row-block max err:", [round(err[i:i + 32].max().item(), 4) for i in range(0, M, 32)])
print("col-block max err:", [round(err[:, i:i + 32].max().item(), 4) for i in range(0, N, 32)])
for K2 in (64, 128, 256):
    x2 = torch.randn(M, K2, device="cuda"); w2 = torch.randn(N, K2, device="cuda") * 0.1
    y2 = ops.linear_f16x3(x2, ops.linear_pack(w2), N)
    print("K=%d max err %.3g" % (K2, (y2.double() - x2.double() @ w2.double().T).abs().max().item()))
