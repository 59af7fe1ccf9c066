"""PCIe-inclusive eval throughput: fp16 bags in pageable host memory -> staging.BagPrefetcher -> fused GA forward,
against the reference's per-slide `.to(device, dtype=float32)` call pattern (Step3_WSI_classification_ACMIL.py:254)."""
import argparse, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd.architecture.transformer import ACMIL_GA
from acmil_amd.staging import BagPrefetcher

ap = argparse.ArgumentParser()
ap.add_argument("--n", type=int, default=50000)
ap.add_argument("--d", type=int, default=512)
ap.add_argument("--slides", type=int, default=64)
ap.add_argument("--distinct", type=int, default=8)
args = ap.parse_args()


class Conf:
    D_feat, D_inner, n_class, n_token = args.d, 256, 2, 5


torch.manual_seed(0)
dev = torch.device("cuda", 0)
m = ACMIL_GA(Conf, n_token=5).to(dev).eval()
bags = [{"input": torch.randn(args.n, args.d).half(), "label": i % 2} for i in range(args.distinct)]
order = [i % args.distinct for i in range(args.slides)]
mb = args.n * args.d * 2 / 1e6

with torch.no_grad():
    for mode in ("reference-style sync fp32 .to()", "sync fp16 .to()", "staged fp16 depth=3"):
        for rep in range(2):
            torch.cuda.synchronize(); t0 = time.perf_counter()
            outs = []
            if mode.startswith("staged"):
                for item in BagPrefetcher(bags, order, dev, depth=3):
                    outs.append(m(item["input"].unsqueeze(0))[1])
            else:
                for i in order:
                    x = bags[i]["input"].to(dev, dtype=torch.float32) if "fp32" in mode else bags[i]["input"].to(dev)
                    outs.append(m(x.unsqueeze(0))[1])
            torch.cuda.synchronize(); dt = time.perf_counter() - t0
        print("%-34s %7.1f slides/s  %6.2f ms/slide  (%.1f GB/s of stored fp16 bytes)" % (mode, args.slides / dt, dt / args.slides * 1e3, mb * args.slides / dt / 1e3))
