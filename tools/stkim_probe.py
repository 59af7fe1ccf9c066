"""Phase stamps of stkim_fused_kernel (csrc/ga_train.hip, -DSTKIM_PROF variant; run through gpurun):
   python tools/stkim_probe.py   -> builds build/variants/libacmil_stkimprof.so, prints the 100 MHz stamps per phase and the launch time."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
out = os.path.join(ROOT, "build", "variants"); os.makedirs(out, exist_ok=True)
lib = os.path.join(out, "libacmil_stkimprof.so")
if "--run" not in sys.argv:
    src = os.path.join(ROOT, "acmil_amd", "csrc")
    obj = os.path.join(out, "ga_train_prof.o")
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"), "-I" + src,
                    "-Wno-unused-value", "-DSTKIM_PROF", "-c", os.path.join(src, "ga_train.hip"), "-o", obj], check=True)
    objs = [os.path.join(src, "build", f) for f in os.listdir(os.path.join(src, "build")) if f.endswith(".o") and f != "ga_train.o"]
    subprocess.run(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs + [obj], check=True)
    env = dict(os.environ, ACMIL_HIP_LIB=lib)
    sys.exit(subprocess.run([sys.executable, os.path.abspath(__file__), "--run"], env=env).returncode)
import ctypes, torch
from acmil_amd import _lib
L = _lib.load()
for N in (10000, 50000):
    K, k, m = 5, 10, 6
    scores = torch.randn(K, N, device="cuda")
    topk = torch.empty(K, k, dtype=torch.int64, device="cuda"); midx = torch.empty(K, m, dtype=torch.int64, device="cuda")
    ws = torch.zeros(L.acmil_stkim_workspace_bytes(N, K, k), dtype=torch.uint8, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    acc = None
    for it in range(30):
        rc = L.acmil_stkim_select_rng(scores.data_ptr(), N, K, k, m, None, 1, it, topk.data_ptr(), midx.data_ptr(), ws.data_ptr(), st)
        assert rc == 0
        torch.cuda.synchronize()
        stamps = ws[64:64 + 7 * 8].view(torch.int64).cpu().tolist()
        d = [(stamps[i] - stamps[0]) * 10 for i in range(7)]      # ns
        if it >= 10:
            acc = d if acc is None else [a + b for a, b in zip(acc, d)]
    print("N=%d  ns from kernel entry (block 0,0): loads %d  extraction %d  ticket %d | last block: start %d  merged(b0) %d  end(b0) %d" %
          tuple([N] + [a // 20 for a in acc[1:]]))
    ref = torch.topk(scores, k, dim=1).indices
    assert torch.equal(ref.cpu(), topk.cpu())
