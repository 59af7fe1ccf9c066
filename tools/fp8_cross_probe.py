"""CPU emulation (numpy / torch-CPU, no GPU): what the split-f16 projection chain would lose if the two cross products
(W_lo x_hi, W_hi x_lo) ran on the FP8 matrix pipe (v_mfma_scale_f32_32x32x64_f8f6f4: twice the f16 rate, BOTH operands 8 bit) while
hi x hi stays f16.  Bag and weights as bench.py draws them (acmil_amd.synthetic), N = 4096 patches.  Prints max |dA_out| against the
fp64 result for: fp32 arithmetic, f16x3 (what the kernel does), hi*hi only, and the FP8 cross-term variants (e4m3 / e5m2 with
power-of-two tensor scales), plus whether the top-10 order per branch survives."""
import os, sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acmil_amd import synthetic as S

torch.manual_seed(0)
N, D, Di, Da, K = 4096, 512, 256, 128, 5
sd = S.ga_state_dict(D, Di, 2, K, Da)
x = S.synthetic_bag(N, D, slide_idx=0)[0].double()
W1 = sd["dimreduction.fc1.weight"].double()
Wv, bv = sd["attention.attention_V.0.weight"].double(), sd["attention.attention_V.0.bias"].double()
Wu, bu = sd["attention.attention_U.0.weight"].double(), sd["attention.attention_U.0.bias"].double()
Ww, bw = sd["attention.attention_weights.weight"].double(), sd["attention.attention_weights.bias"].double()


def split16(a):
    hi = a.float().half()
    lo = (a.float() - hi.float()).half()
    return hi.double(), lo.double()


def q8(a, fmt):
    """round to an 8-bit float with a power-of-two scale that puts max|a| near the top of the format's range"""
    t = a.float()
    m = t.abs().max().item()
    if m == 0:
        return a
    top = 448.0 if fmt == torch.float8_e4m3fn else 57344.0
    s = 2.0 ** np.floor(np.log2(top / m))
    return ((t * s).to(fmt).float() / s).double()


def scores(matmul1, matmul2):
    h = torch.relu(matmul1(x, W1))
    g = matmul2(h, torch.cat([Wv, Wu])) + torch.cat([bv, bu])
    gate = torch.tanh(g[:, :Da]) * torch.sigmoid(g[:, Da:])
    return (gate @ Ww.T + bw).T           # [K, N]


def mm_exact(a, w):
    return a @ w.T


def mm_fp32(a, w):
    return (a.float() @ w.float().T).double()


def mm_split(cross):
    def f(a, w):
        ah, al = split16(a)
        wh, wl = split16(w)
        main = ah @ wh.T                      # f16 x f16 products are exact in fp32; accumulation error ignored here (fp64 sums)
        if cross == "none":
            return main
        if cross == "f16":
            return main + ah @ wl.T + al @ wh.T
        fmt = torch.float8_e4m3fn if cross == "e4m3" else torch.float8_e5m2
        return main + q8(ah, fmt) @ q8(wl, fmt).T + q8(al, fmt) @ q8(wh, fmt).T
    return f


ref = scores(mm_exact, mm_exact)
top_ref = torch.topk(ref, 10, dim=1).indices
for name, m in (("fp32 arithmetic", mm_fp32), ("f16x3 (kernel)", mm_split("f16")), ("hi*hi only", mm_split("none")),
                ("cross terms on fp8 e4m3", mm_split("e4m3")), ("cross terms on fp8 e5m2", mm_split("e5m2"))):
    a = scores(m, m)
    same = bool((torch.topk(a, 10, dim=1).indices == top_ref).all())
    print("%-26s max |dA_out| %.2e   top-10 order identical: %s" % (name, (a - ref).abs().max().item(), same))
