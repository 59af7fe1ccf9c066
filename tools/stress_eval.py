import os, sys
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from acmil_amd import ops, synthetic as S
dev = torch.device("cuda")
bad = 0
for (D, Di, dt) in ((384, 128, torch.bfloat16), (384, 128, torch.float16), (512, 256, torch.float32)):
    sd = {k: v.to(dev) for k, v in S.ga_state_dict(D, Di, 2, 5).items()}
    packed, dims = ops.ga_pack_weights(sd["dimreduction.fc1.weight"], sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"],
                                       sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"],
                                       sd["attention.attention_weights.weight"], sd["attention.attention_weights.bias"],
                                       [sd["classifier.%d.fc.weight" % i] for i in range(5)], [sd["classifier.%d.fc.bias" % i] for i in range(5)],
                                       sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"], "f16x3")
    ns = [50000, 1, 257, 33000, 129, 7777, 50000, 12345] * 2
    xs = [S.synthetic_bag(n, D, slide_idx=i)[0].to(dt).to(dev) for i, n in enumerate(ns)]
    ref = None
    for r in range(60):
        out = ops.ga_forward_batch(xs, packed, dims, "f16x3", want_bag_feat=True)
        cur = [out["sub_preds"].clone(), out["slide_pred"].clone(), out["bag_feat"].clone()] + [a.clone() for a in out["A_out"]]
        if ref is None: ref = cur
        else:
            for i, (a, b) in enumerate(zip(ref, cur)):
                if not torch.equal(a, b):
                    bad += 1; print("MISMATCH", D, Di, dt, r, i); break
    torch.cuda.synchronize()
    print(D, Di, dt, "60 repeats checked")
print("STRESS_EVAL", "FAIL" if bad else "OK")
sys.exit(1 if bad else 0)
