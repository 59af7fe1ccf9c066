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

# ---- further paths beside the TransMIL-forward disturber (its short Moore-Penrose workgroups are what delayed single waves of lin_kernel)
from acmil_amd.architecture.transformer import ACMIL_MHA
from acmil_amd.architecture.transMIL import TransMIL
class ConfW: D_feat, D_inner, n_class, n_token = 1024, 512, 2, 5
class ConfM: D_feat, D_inner, n_class, n_token = 512, 256, 2, 5
class ConfT: D_feat, D_inner, n_class = 512, 256, 2
torch.manual_seed(0)
wide = ACMIL_GA(ConfW, n_token=5, n_masked_patch=10, mask_drop=0.6).cuda()
xw = O.synthetic_bag(20000, 1024, slide_idx=3).cuda()
mha = ACMIL_MHA(ConfM, n_token=5, n_masked_patch=10, mask_drop=0.6).cuda().eval()
xm = O.synthetic_bag(20000, 512, slide_idx=4).cuda()
tmt = TransMIL(ConfT).cuda().train()
for mod in tmt.modules():
    if isinstance(mod, torch.nn.Dropout): mod.p = 0.0
xtt = O.synthetic_bag(4000, 512, slide_idx=5).cuda()
uw = torch.rand(5, 10, generator=torch.Generator().manual_seed(3)).cuda(); yw = torch.tensor([1], device="cuda")
def wide_eval():
    wide.eval()
    with torch.no_grad(): return [t.clone() for t in wide(xw)]
def wide_train():
    wide.train()
    losses, _ = wide.train_step(xw, yw, uniforms=uw)
    return [losses.clone()] + [p.grad.clone() for p in wide.parameters()]
def mha_eval():
    with torch.no_grad(): return [t.clone() for t in mha(xm)]
def tm_train():
    tmt.zero_grad(set_to_none=True)
    lg = tmt(xtt); torch.nn.functional.cross_entropy(lg, yw).backward()
    return [lg.detach().clone()] + [p.grad.clone() for p in tmt.parameters() if p.grad is not None]
for name, fn, n in (("wide GA eval", wide_eval, 60), ("wide GA training step", wide_train, 40), ("ACMIL_MHA eval", mha_eval, 60), ("TransMIL training step", tm_train, 15)):
    r0 = fn(); torch.cuda.synchronize()
    got = []
    def loop():
        st = torch.cuda.Stream()
        with torch.cuda.stream(st):
            for _ in range(n): got.append(fn())
        st.synchronize()
    ts = [threading.Thread(target=loop), threading.Thread(target=tm_loop, args=(1, 40))]
    [t.start() for t in ts]; [t.join() for t in ts]; torch.cuda.synchronize()
    print("%s beside TransMIL forwards, %d: deviating %d" % (name, n, sum(1 for o in got if not all(torch.equal(a, b) for a, b in zip(o, r0)))))
