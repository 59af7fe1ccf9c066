"""Run-to-run stress of the lo-product skip (ga_fwd3_kernel's asm-embedded branch, lin_kernel / lin64_kernel) and of the grouped eval (run through
gpurun): UNI / CLIP-L / GigaPath models on fp16-valued fp32 bags, bags with a few genuine fp32 rows and fp16-stored bags, 150 rounds each,
every result bit-identical to the first round and the fp16-valued ones to the fp16-stored ones.  Prints the number of deviating calls."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import synthetic as S
from acmil_amd.architecture.transformer import ACMIL_GA

dev = torch.device("cuda", 0)
bad = 0
for (D, Di) in ((1024, 512), (768, 384), (1536, 768)):
    class Conf:
        D_feat, D_inner, n_class, n_token = D, Di, 2, 5
    model = ACMIL_GA(Conf, n_token=5, n_masked_patch=10, mask_drop=0.6)
    model.load_state_dict(S.ga_state_dict(D, Di, 2, 5, seed=D))
    model = model.to(dev).eval()
    g = torch.Generator().manual_seed(D)
    ns = [5000, 129, 20000, 777]
    x16 = [S.synthetic_bag(n, D, slide_idx=i)[0].half() for i, n in enumerate(ns)]
    xe = [x.float().to(dev) for x in x16]
    xm = []
    for x in x16:
        y = x.float()
        rows = torch.randint(0, y.shape[0], (max(1, y.shape[0] // 200),), generator=g)
        y[rows] += torch.randn(len(rows), D, generator=g) * 1e-4
        xm.append(y.to(dev))
    xh = [x.to(dev) for x in x16]
    with torch.no_grad():
        def run():
            outs = []
            for bags in (xe, xm, xh):
                for b in bags:
                    outs.append(model(b.unsqueeze(0)))
                outs += model.forward_group(torch.cat(bags, 0), ns)
            return outs
        ref = run()
        n_each = len(ns) * 2
        for (a, b) in zip(ref[:n_each], ref[2 * n_each:3 * n_each]):      # fp16-valued == fp16-stored
            if not (torch.equal(a[2], b[2]) and torch.equal(a[0], b[0])):
                bad += 1
        for (a, b) in zip(ref[:len(ns)], ref[len(ns):n_each]):              # per slide == grouped
            if not (torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])):
                bad += 1
        for it in range(150):
            out = run()
            for a, b in zip(out, ref):
                if not (torch.equal(a[2], b[2]) and torch.equal(a[0], b[0]) and torch.equal(a[1], b[1])):
                    bad += 1
    print("D=%d D_inner=%d: deviating calls so far %d, range fallbacks %d" % (D, Di, bad, model.range_fallbacks))
print("STRESS_LOSKIP deviating calls:", bad)
