"""A/B probe of the fused GA forward kernel variants on the GPU box (run through gpurun).

Each variant = a set of environment variables (ACMIL_GA_KERNEL, ACMIL_GA_WAVES, ACMIL_HIP_LIB ...) evaluated in its OWN
subprocess (the library reads them once), on the north-star shape: 16 resident fp32 bags of 50 000 x 512, K=5, C=2.
Per variant: kernel-level time of the batched launch (events on the launch stream, interleaved rounds, median and min),
the single-bag time, and the outputs of bag 0 for a cross-variant comparison against the first variant.

  python tools/probe_ga.py                      # default variant list
  python tools/probe_ga.py --variants "v1w4:ACMIL_GA_KERNEL=1,ACMIL_GA_WAVES=4;v2w4:ACMIL_GA_KERNEL=2,ACMIL_GA_WAVES=4"
"""
import argparse, json, os, subprocess, sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

DEFAULT = ("v1w4:ACMIL_GA_KERNEL=1,ACMIL_GA_WAVES=4;v1w8:ACMIL_GA_KERNEL=1,ACMIL_GA_WAVES=8;"
           "v2w4:ACMIL_GA_KERNEL=2,ACMIL_GA_WAVES=4;v2w8:ACMIL_GA_KERNEL=2,ACMIL_GA_WAVES=8")


def child(args):
    import torch
    from acmil_amd import ops
    from acmil_amd import synthetic as SY
    dev = "cuda"
    K, C = args.k, args.c
    sd = {k: v.to(dev) for k, v in SY.ga_state_dict(args.d, args.di, C, K).items()}
    packed, dims = ops.ga_pack_weights(
        sd["dimreduction.fc1.weight"], sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"],
        sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"], sd["attention.attention_weights.weight"],
        sd["attention.attention_weights.bias"], [sd["classifier.%d.fc.weight" % i] for i in range(K)],
        [sd["classifier.%d.fc.bias" % i] for i in range(K)], sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"], args.mode)
    g = torch.Generator(device="cpu").manual_seed(1234)
    dt = getattr(torch, args.xdtype)
    bags = []
    for b in range(args.bags):
        n = args.n - (37 * b if args.ragged else 0)
        bags.append(torch.randn(n, args.d, generator=g).to(dt).to(dev))
    torch.cuda.synchronize()

    def run_batch():
        return ops.ga_forward_batch(bags[:args.batch], packed, dims, args.mode, want_scores=True)

    def run_one(i):
        return ops.ga_forward(bags[i % args.bags], packed, dims, args.mode)

    out = run_batch()
    torch.cuda.synchronize()
    res = {"name": args.name}
    # save bag-0 outputs for the parent's cross-check
    torch.save({"A0": out["A_out"][0].cpu(), "sub": out["sub_preds"].cpu(), "slide": out["slide_pred"].cpu(),
                "Alast": out["A_out"][-1].cpu()}, args.save)
    for tag, fn, per in (("batch", run_batch, args.batch), ("one", lambda: run_one(0), 1)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        ts = []
        for r in range(args.rounds):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for i in range(args.iters):
                fn() if tag == "batch" else run_one(i)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) / args.iters * 1e3)
        ts.sort()
        res[tag + "_us_med"] = round(ts[len(ts) // 2], 1)
        res[tag + "_us_min"] = round(ts[0], 1)
        res[tag + "_us_per_slide"] = round(ts[len(ts) // 2] / per, 2)
    if os.environ.get("GA_PROF"):
        A0 = out["A_out"][0][0]
        n32 = (A0.numel() // 32) * 32
        pv = A0[:n32].view(-1, 32)[:, :8].double()
        res["prof_mean"] = [round(float(v)) for v in pv.mean(0)]
        res["prof_names"] = ["total", "g1_vm", "g1_bar", "softmax", "pool", "g1", "g2", "combine"]
        import numpy as np
        rows = []
        for b in range(len(out["A_out"])):
            Ab = out["A_out"][b][0]
            nb = (Ab.numel() // 32) * 32
            rows.append(Ab[:nb].view(-1, 32)[:, :16].cpu().numpy())
        np.save(args.save.replace(".pt", "_prof.npy"), np.concatenate(rows, 0))
    print("PROBE " + json.dumps(res), flush=True)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--variants", default=DEFAULT)
    ap.add_argument("--n", type=int, default=50000)
    ap.add_argument("--d", type=int, default=512)
    ap.add_argument("--di", type=int, default=256)
    ap.add_argument("--k", type=int, default=5)
    ap.add_argument("--c", type=int, default=2)
    ap.add_argument("--bags", type=int, default=16)
    ap.add_argument("--batch", type=int, default=16)
    ap.add_argument("--iters", type=int, default=10)
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--mode", default="f16x3")
    ap.add_argument("--xdtype", default="float32")
    ap.add_argument("--ragged", action="store_true")
    ap.add_argument("--child", action="store_true")
    ap.add_argument("--name", default="")
    ap.add_argument("--save", default="")
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "probe"))
    args = ap.parse_args()
    if args.child:
        return child(args)
    os.makedirs(args.out, exist_ok=True)
    import torch
    results, ref = [], None
    for spec in args.variants.split(";"):
        name, _, envs = spec.partition(":")
        env = dict(os.environ)
        for kv in filter(None, envs.split(",")):
            k, _, v = kv.partition("=")
            env[k] = v
        save = os.path.join(args.out, name + ".pt")
        cmd = [sys.executable, os.path.abspath(__file__), "--child", "--name", name, "--save", save]
        for k in ("n", "d", "di", "k", "c", "bags", "batch", "iters", "rounds", "mode", "xdtype"):
            cmd += ["--" + k, str(getattr(args, k))]
        if args.ragged:
            cmd.append("--ragged")
        r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=600)
        line = [l for l in r.stdout.splitlines() if l.startswith("PROBE ")]
        if r.returncode != 0 or not line:
            print("VARIANT %s FAILED rc=%d\n%s" % (name, r.returncode, r.stdout[-1500:]), flush=True)
            continue
        res = json.loads(line[0][6:])
        if "prof_mean" in res:
            print("PROF %s %s" % (name, dict(zip(res["prof_names"], res["prof_mean"]))), flush=True)
        cur = torch.load(save)
        if ref is None:
            ref = cur
        res["max_dA_vs_first"] = max(float((cur[k] - ref[k]).abs().max()) for k in ("A0", "Alast"))
        res["max_dlogit_vs_first"] = max(float((cur[k] - ref[k]).abs().max()) for k in ("sub", "slide"))
        results.append(res)
        print(json.dumps(res), flush=True)
        os.remove(save) if ref is not cur else None
    with open(os.path.join(args.out, "probe.json"), "w") as fh:
        json.dump(results, fh, indent=1)


if __name__ == "__main__":
    main()
