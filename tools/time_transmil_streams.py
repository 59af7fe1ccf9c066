"""TransMIL forwards (cfg4: N = 100 000, D = 768) on one stream vs on two host threads / two streams (run through gpurun): slides per second."""
import sys, threading, time, torch, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from acmil_amd import ops
from acmil_amd import synthetic as S
d, di, n = 768, 384, 100000
sd = {k: v.cuda() for k, v in S.transmil_state_dict(d, di, 2, seed=1).items()}
bags = [torch.randn(n, d, device="cuda") for _ in range(2)]
def loop(i, iters):
    st = torch.cuda.Stream()
    with torch.cuda.stream(st):
        for _ in range(iters): ops.transmil_forward(bags[i], sd, 2)
    st.synchronize()
loop(0, 5); torch.cuda.synchronize()
t0 = time.perf_counter(); loop(0, 60); t1 = time.perf_counter() - t0
print("one stream : %.1f slides/s (%.3f ms per slide)" % (60 / t1, t1 / 60 * 1e3))
ts = [threading.Thread(target=loop, args=(i, 30)) for i in range(2)]
t0 = time.perf_counter(); [t.start() for t in ts]; [t.join() for t in ts]; torch.cuda.synchronize(); t2 = time.perf_counter() - t0
print("two streams: %.1f slides/s (%.3f ms per slide)" % (60 / t2, t2 / 60 * 1e3))
