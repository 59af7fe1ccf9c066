"""Timing of the packed-weight Linear kernels (lin_kernel vs lin64_kernel: ACMIL_LIN64=0 / 1 per subprocess) on the TransMIL cfg4 and
wide-GA projection shapes, with a correctness check against fp64 on a row sample.  Run on the GPU box:  python tools/time_linear.py"""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

SHAPES = [(100000, 768, 384), (100224, 384, 1152), (100224, 384, 384), (50000, 1024, 512), (50000, 1536, 768), (50000, 768, 384), (3000, 512, 256)]


def child():
    import torch
    from acmil_amd import ops
    for (m, k, n) in SHAPES:
        x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.03
        b = torch.randn(n, device="cuda")
        packed = ops.linear_pack(w)
        y = torch.empty(m, n, device="cuda")
        fn = lambda: ops.linear_f16x3(x, packed, n, bias=b, relu=True, out=y)
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / 20 * 1e3
        rows = torch.cat([torch.arange(0, min(m, 300)), torch.arange(max(0, m - 300), m), torch.randint(0, m, (400,))]).cuda()
        ref = torch.relu(x[rows].double() @ w.double().T + b.double())
        scale = (x[rows].double().abs() @ w.double().abs().T).max().item()
        err = (y[rows].double() - ref).abs().max().item() / scale
        print("LIN64=%s M=%d K=%d N=%d: %.1f us  %.0f TF algorithmic (%.3f of 2.5 PF executed)  rel err %.1e" % (
            os.environ.get("ACMIL_LIN64", "auto"), m, k, n, us, 2.0 * m * k * n / us / 1e6, 6.0 * m * k * n / us / 1e6 / 2500.0, err), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        ab = os.path.join(ROOT, "acmil_amd", "libacmil_hip_ab.so")      # the knobs exist in the A/B build only
        for env in ({"ACMIL_LIN64": "0"}, {"ACMIL_LIN64": "1"}, {"ACMIL_LIN_WAVES": "8"}, {"ACMIL_LIN64": "0"}, {"ACMIL_LIN_WAVES": "8"}):
            print("==", env, flush=True)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=dict(os.environ, ACMIL_HIP_LIB=ab, **env))
