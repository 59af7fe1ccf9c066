"""Single-bag fused forward: launch time vs bag size (events on the stream, ops.ga_forward incl. merge + heads) -- run through gpurun.
ACMIL_GA2_WAVES=4|8 forces the tile geometry."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import ops, synthetic as S

dev = torch.device("cuda")
sd = {k: v.to(dev) for k, v in S.ga_state_dict(512, 256, 2, 5).items()}
packed, dims = ops.ga_pack_weights(sd["dimreduction.fc1.weight"], sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"],
                                   sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"],
                                   sd["attention.attention_weights.weight"], sd["attention.attention_weights.bias"],
                                   [sd["classifier.%d.fc.weight" % i] for i in range(5)], [sd["classifier.%d.fc.bias" % i] for i in range(5)],
                                   sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"], "f16x3")
for n in (128, 1024, 4096, 8192, 16384, 32768, 40000, 50000, 57344, 65536, 100000):
    bags = [torch.randn(n, 512, device=dev) for _ in range(4)]
    for i in range(20):
        ops.ga_forward(bags[i % 4], packed, dims, "f16x3")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(200):
        ops.ga_forward(bags[i % 4], packed, dims, "f16x3")
    e1.record()
    torch.cuda.synchronize()
    print("N %6d  tiles128 %4d  %.1f us per forward" % (n, (n + 127) // 128, e0.elapsed_time(e1) * 1e3 / 200))
