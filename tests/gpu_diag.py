"""Ad-hoc GPU diagnostic (not a pytest file): per-mode error table of the fused forward vs the oracle."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from oracle import ga_oracle as O
from acmil_amd.architecture.transformer import ACMIL_GA


def run(n, d, di, k, c, precision, xdtype=torch.float32):
    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k
    sd = O.default_state_dict(d, di, c, k)
    m = ACMIL_GA(Conf, n_token=k, precision=precision)
    m.load_state_dict(sd); m = m.cuda().eval()
    x = O.synthetic_bag(n, d, 1).to(xdtype)
    ref = O.acmil_ga_forward(x.float(), sd, n_token=k)
    ref64 = O.acmil_ga_forward(x.double(), {kk: v.double() for kk, v in sd.items()}, n_token=k)
    with torch.no_grad():
        sub, slide, a = m(x.cuda())
    torch.cuda.synchronize()
    ea = (a.cpu() - ref["A_out"]).abs().max().item()
    ea64 = (a.cpu().double() - ref64["A_out"]).abs().max().item()
    er64 = (ref["A_out"].double() - ref64["A_out"]).abs().max().item()
    es = (sub.cpu() - ref["sub_preds"]).abs().max().item()
    eb = (slide.cpu() - ref["slide_pred"]).abs().max().item()
    t10 = torch.topk(a[0].cpu(), min(10, n), dim=-1).indices
    r10 = torch.topk(ref["A_out"][0], min(10, n), dim=-1).indices
    print("N=%6d D=%d Di=%d K=%d C=%d %-6s %-8s dA=%.2e (vs fp64 %.2e; ref-vs-fp64 %.2e) dsub=%.2e dslide=%.2e top10_equal=%s"
          % (n, d, di, k, c, precision, str(xdtype).split('.')[-1], ea, ea64, er64, es, eb, bool(torch.equal(t10, r10))), flush=True)


if __name__ == "__main__":
    print(torch.cuda.get_device_name(0))
    for prec in ["fp32", "f16x3", "f16"]:
        for (n, d, di, k, c) in [(1, 512, 256, 5, 2), (33, 512, 256, 5, 2), (257, 512, 256, 5, 2), (1000, 512, 256, 1, 2),
                                 (1000, 384, 128, 5, 7), (10000, 512, 256, 5, 2), (50000, 512, 256, 5, 2)]:
            try:
                run(n, d, di, k, c, prec)
            except Exception as e:
                print("FAIL", prec, n, d, di, k, c, repr(e), flush=True)
    for xdt in [torch.float16, torch.bfloat16]:
        for prec in ["fp32", "f16x3"]:
            run(5000, 512, 256, 5, 2, prec, xdt)
    # quick timing
    from acmil_amd import ops
    for prec in ["fp32", "f16x3", "f16"]:
        class Conf:
            D_feat, D_inner, n_class, n_token = 512, 256, 2, 5
        m = ACMIL_GA(Conf, n_token=5, precision=prec).cuda().eval()
        xs = [torch.randn(1, 50000, 512, device="cuda") for _ in range(8)]
        with torch.no_grad():
            for i in range(5): m(xs[i % 8])
            torch.cuda.synchronize(); t0 = time.time()
            for i in range(50): m(xs[i % 8])
            torch.cuda.synchronize(); dt = (time.time() - t0) / 50
        print("timing %-6s N=50000: %.1f us/slide" % (prec, dt * 1e6), flush=True)
