"""Kernel-level timing of the fused GA forward at a chosen shape (no module, no host sync per call):
   python tools/time_v3.py D Di [N] [batch] [dtype]     -- us per launch (torch events), fused launch + merge + heads."""
import sys
import torch
sys.path.insert(0, ".")
from acmil_amd import ops
from acmil_amd import synthetic as S

d, di = int(sys.argv[1]), int(sys.argv[2])
n = int(sys.argv[3]) if len(sys.argv) > 3 else 50000
batch = int(sys.argv[4]) if len(sys.argv) > 4 else 1
dt = {"f32": torch.float32, "f16": torch.float16, "bf16": torch.bfloat16}[sys.argv[5] if len(sys.argv) > 5 else "f32"]
k, c = 5, 2
sd = {a: v.cuda() for a, v in S.ga_state_dict(d, di, c, k, seed=0).items()}
packed, dims = ops.ga_pack_weights(sd['dimreduction.fc1.weight'], sd['attention.attention_V.0.weight'], sd['attention.attention_V.0.bias'],
    sd['attention.attention_U.0.weight'], sd['attention.attention_U.0.bias'], sd['attention.attention_weights.weight'],
    sd['attention.attention_weights.bias'], [sd['classifier.%d.fc.weight' % i] for i in range(k)],
    [sd['classifier.%d.fc.bias' % i] for i in range(k)], sd['Slide_classifier.fc.weight'], sd['Slide_classifier.fc.bias'], 'f16x3')
nb = max(8, batch)
bags = [torch.randn(n, d, generator=torch.Generator().manual_seed(i)).to(dt).cuda() for i in range(nb)]
def step(i):
    if batch == 1:
        ops.ga_forward(bags[i % nb], packed, dims, 'f16x3')
    else:
        ops.ga_forward_batch([bags[(i + j) % nb] for j in range(batch)], packed, dims, 'f16x3')
for i in range(10): step(i)
torch.cuda.synchronize()
reps = 50
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for i in range(reps): step(i)
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1000 / reps
flops = 2.0 * n * (d * di + 2 * di * 128 + 128 * k + k * di) * batch
print("D=%d Di=%d N=%d batch=%d %s: %.1f us per step (%.1f us/slide), %.0f TF algorithmic, executed_frac %.3f" % (d, di, n, batch, sys.argv[5] if len(sys.argv) > 5 else "f32", us, us / batch, flops / us / 1e6, 3 * flops / us / 1e6 / 2500))
