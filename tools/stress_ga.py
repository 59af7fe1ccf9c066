"""Stress of the GA paths under co-running launches (run through gpurun): eval forwards and training steps (forward + STKIM + losses + backward,
injected uniforms) on two host threads / two streams beside each other and beside TransMIL forwards; every result must be bit-identical to the
single-threaded one.  Prints the number of deviating calls."""
import sys, threading, torch, os, copy
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acmil_amd import ops
from acmil_amd import synthetic as S
from acmil_amd.architecture.transformer import ACMIL_GA
from oracle import ga_oracle as O
class Conf: D_feat, D_inner, n_class, n_token = 512, 256, 7, 5
torch.manual_seed(0)
base = ACMIL_GA(Conf, n_token=5, n_masked_patch=10, mask_drop=0.6).cuda()
models = [copy.deepcopy(base).train() for _ in range(2)]
bags = [O.synthetic_bag(n, 512, slide_idx=i).half().cuda() for i, n in enumerate((10000, 23000))]
ys = [torch.tensor([1], device="cuda"), torch.tensor([3], device="cuda")]
us = [torch.rand(5, 10, generator=torch.Generator().manual_seed(7 + i)).cuda() for i in range(2)]
def step(i):
    m = models[i]
    losses, _ = m.train_step(bags[i], ys[i], uniforms=us[i])
    return [losses.clone()] + [p.grad.clone() for p in m.parameters()]
ref = [step(i) for i in range(2)]
torch.cuda.synchronize()
d, di = 768, 384
sd = {k: v.cuda() for k, v in S.transmil_state_dict(d, di, 2, seed=7).items()}
xt = torch.randn(9000, d, generator=torch.Generator().manual_seed(9000)).cuda()
res = {}
def train_loop(i, n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        outs = [step(i) for _ in range(n)]
    st.synchronize(); res[i] = outs
def tm_loop(i, n):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(n): ops.transmil_forward(xt, sd, 2)
    st.synchronize()
def bad(i): return sum(1 for o in res[i] if not all(torch.equal(a, b) for a, b in zip(o, ref[i])))
ts = [threading.Thread(target=train_loop, args=(i, 60)) for i in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]; torch.cuda.synchronize()
print("training steps, two threads / two streams, 60 each: deviating", [bad(0), bad(1)])
ts = [threading.Thread(target=train_loop, args=(0, 60)), threading.Thread(target=tm_loop, args=(1, 40))]
[t.start() for t in ts]; [t.join() for t in ts]; torch.cuda.synchronize()
print("training steps beside TransMIL forwards, 60: deviating", bad(0))
