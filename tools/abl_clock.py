"""Launch time + shader clock + socket power of the batched GA forward for ONE library build (ACMIL_HIP_LIB selects it).

Run once per variant (tools/build_variants.sh) in its own process; prints one ABLCLK json line: us per 16-bag launch (events on
the launch stream over ~3 s of back-to-back launches), effective clock and power sampled with rocm-smi while the loop runs.
Timing-only variants (GA2_ABL) produce WRONG results; this tool never checks outputs.
"""
import json, os, re, subprocess, sys, threading, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from acmil_amd import ops
from acmil_amd import synthetic as SY

name = sys.argv[1] if len(sys.argv) > 1 else "base"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 16
xdt = getattr(torch, sys.argv[3]) if len(sys.argv) > 3 else torch.float32
sd = {k: v.cuda() for k, v in SY.ga_state_dict(512, 256, 2, 5).items()}
packed, dims = ops.ga_pack_weights(sd["dimreduction.fc1.weight"], sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"],
    sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"], sd["attention.attention_weights.weight"],
    sd["attention.attention_weights.bias"], [sd["classifier.%d.fc.weight" % i] for i in range(5)],
    [sd["classifier.%d.fc.bias" % i] for i in range(5)], sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"], "f16x3")
bags = [torch.randn(50000, 512, device="cuda").to(xdt) for _ in range(16)]
samples, stop = [], False


def smi():
    r = subprocess.run(["rocm-smi", "--showclocks", "--showpower", "--json"], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL, text=True)
    try:
        return json.loads(r.stdout)
    except Exception:
        return {}


def sampler():
    while not stop:
        samples.append(smi())
        time.sleep(0.15)


def run(i):
    if batch == 1:
        ops.ga_forward(bags[i % 16], packed, dims, "f16x3")
    else:
        ops.ga_forward_batch(bags[:batch], packed, dims, "f16x3")


for i in range(20):
    run(i)
torch.cuda.synchronize()
t = threading.Thread(target=sampler); t.start()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = 0
t0 = time.time()
e0.record()
while time.time() - t0 < 3.0:
    for i in range(50):
        run(n + i)
    n += 50
    torch.cuda.synchronize()
e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) * 1e3 / n
stop = True; t.join()
sclk, pwr = [], []
for s in samples[2:]:
    for card, d in s.items():
        for k, v in d.items():
            if "sclk" in k.lower():
                m = re.search(r"(\d+)\s*Mhz", str(v), re.I)
                if m: sclk.append(int(m.group(1)))
            if "power" in k.lower() and "W" in k:
                try: pwr.append(float(v))
                except Exception: pass
        break
res = {"name": name, "batch": batch, "us_per_launch": round(us, 1), "n": n,
       "sclk_mhz_mean": round(sum(sclk) / len(sclk)) if sclk else None, "sclk_min": min(sclk) if sclk else None, "sclk_max": max(sclk) if sclk else None,
       "power_w_mean": round(sum(pwr) / len(pwr)) if pwr else None, "power_max": max(pwr) if pwr else None, "nsamples": len(samples)}
if not sclk or not pwr:
    res["raw"] = samples[3] if len(samples) > 3 else samples[-1:]
print("ABLCLK " + json.dumps(res), flush=True)
