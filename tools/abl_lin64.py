"""Ablation timing of lin64_kernel (tools/build_lin_variants.sh variants; timing only, results are wrong by construction)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
SHAPES = [(100224, 384, 1152), (50000, 1024, 512)]


def child():
    import torch
    from acmil_amd import ops
    out = []
    for (m, k, n) in SHAPES:
        x = torch.randn(m, k, device="cuda"); w = torch.randn(n, k, device="cuda") * 0.03
        packed = ops.linear_pack(w)
        y = torch.empty(m, n, device="cuda")
        fn = lambda: ops.linear_f16x3(x, packed, n, out=y)
        for _ in range(10): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50): fn()
        e1.record(); torch.cuda.synchronize()
        out.append("%7.1f us" % (e0.elapsed_time(e1) / 50 * 1e3))
    print("%-10s %s" % (os.environ.get("ABL_NAME", "base"), "   ".join(out)), flush=True)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        print("variant    " + "   ".join("M=%d K=%d N=%d" % s for s in SHAPES))
        names = ["base"] + sys.argv[1:]
        for nm in names:
            env = dict(os.environ, ACMIL_LIN64="1", ABL_NAME=nm)
            if nm == "lin32":
                env["ACMIL_LIN64"] = "0"
            elif nm.startswith("l32_"):      # a variant build of linear.hip run through lin_kernel
                env["ACMIL_LIN64"] = "0"
                env["ACMIL_HIP_LIB"] = os.path.join(ROOT, "build", "variants", "libacmil_%s.so" % nm[4:])
            elif nm != "base":
                env["ACMIL_HIP_LIB"] = os.path.join(ROOT, "build", "variants", "libacmil_%s.so" % nm)
            subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=env)
