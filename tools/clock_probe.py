"""Sample shader clock / power (rocm-smi) while the batched GA forward runs back to back for a few seconds."""
import os, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import ops
from acmil_amd import synthetic as SY
sd = {k: v.cuda() for k, v in SY.ga_state_dict(512, 256, 2, 5).items()}
packed, dims = ops.ga_pack_weights(sd["dimreduction.fc1.weight"], sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"],
    sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"], sd["attention.attention_weights.weight"],
    sd["attention.attention_weights.bias"], [sd["classifier.%d.fc.weight" % i] for i in range(5)],
    [sd["classifier.%d.fc.bias" % i] for i in range(5)], sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"], "f16x3")
bags = [torch.randn(50000, 512, device="cuda") for _ in range(16)]
samples = []
stop = False
def sampler():
    while not stop:
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
        samples.append(r.stdout.strip().splitlines()[-1] if r.stdout.strip() else "")
        time.sleep(0.2)
for mode in (sys.argv[1:] or ["run"]):
    t = threading.Thread(target=sampler); stop = False; samples.clear(); t.start()
    t0 = time.time(); n = 0
    while time.time() - t0 < 4.0:
        for _ in range(20):
            ops.ga_forward_batch(bags, packed, dims, "f16x3")
        torch.cuda.synchronize(); n += 20
    el = time.time() - t0
    stop = True; t.join()
    print("CLOCK %s: %.1f us/launch over %.1f s" % (mode, el / n * 1e6, el))
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--csv"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    print(r.stdout.strip().splitlines()[0][:300])
    for s in samples[2:12]:
        print("  ", s[:300])
