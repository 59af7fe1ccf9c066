"""Stress of the TransMIL forward (run through gpurun): many back-to-back forwards of alternating bags on one stream and on two host
threads / two streams; every result must be bit-identical to the first one of its bag.  Prints the number of deviating forwards."""
import sys, threading, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acmil_amd import ops
from acmil_amd import synthetic as S
d, di = 768, 384
sd = {k: v.cuda() for k, v in S.transmil_state_dict(d, di, 2, seed=7).items()}
bags = [torch.randn(n, d, generator=torch.Generator().manual_seed(n)).cuda() for n in (3000, 9000, 40000)]
ref = [ops.transmil_forward(x, sd, 2)["logits"].clone() for x in bags]
torch.cuda.synchronize()
outs = [(i % 3, ops.transmil_forward(bags[i % 3], sd, 2)["logits"]) for i in range(300)]
torch.cuda.synchronize()
print("one stream, 300 forwards of three alternating bags: deviating", sum(1 for i, o in outs if not torch.equal(o, ref[i])))
res = {}
def work(t):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        o = [(i % 3, ops.transmil_forward(bags[(i + t) % 3], sd, 2)["logits"]) for i in range(100)]
    st.synchronize(); res[t] = [(((i + t) % 3), x) for (i, x) in o]
ts = [threading.Thread(target=work, args=(t,)) for t in range(2)]
[t.start() for t in ts]; [t.join() for t in ts]
torch.cuda.synchronize()
print("two threads / two streams, 100 forwards each: deviating", [sum(1 for i, o in res[t] if not torch.equal(o, ref[i])) for t in range(2)])
