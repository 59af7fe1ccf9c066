#!/usr/bin/env python
"""bench.py -- headline benchmark of the ACMIL gated-attention aggregation hot path on MI355X.

Contract (driver):  python bench.py --gpus N --steps K --warmup W
  N > 1 is launched as `python -m torch.distributed.run --nproc-per-node N ... bench.py --gpus N ...`
  (one rank per GPU; RANK / LOCAL_RANK / WORLD_SIZE / MASTER_* from the environment).

Workload (BASELINE.json metric): ACMIL-ga eval forward, N=50 000 patches, D=512, D_inner=256, K=5 branches,
C=2 classes, fp32 bag resident in HBM, weights = torch default nn.Linear init (random), synthetic randn bags.
A "step" is one pass of the hot path over one batch of --batch slides (default 16 = the most one launch takes; every bag is still an independent
B=1 problem exactly as in the reference, the batch only shares one launch: acmil_ga_forward_batch): weight stream
already packed, fused forward + merge + heads through the C ABI, 16 distinct bags rotated so neither L2 nor the
256 MB Infinity Cache holds the working set.  The strictly per-slide (B=1 call) latency is reported next to it.  Slides shard across GPUs with no data-path collective
(eval forward: pure replicas over disjoint slides) -> "scaling": "weak".

One JSON line on rank 0: slides/s (whole job) + roofline of the dominant kernel (ga_fwd_kernel, timed with
events on the launch stream) + the CPU baseline (the oracle = port of the reference's PyTorch-CPU forward,
timed on this box's host cores on a bounded sample).
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import torch  # noqa: E402

N_PATCH, D_FEAT, D_INNER, N_TOKEN, N_CLASS, D_ATTN = 50000, 512, 256, 5, 2, 128
N_BAGS = 16      # resident bags (grown to the batch size in main)


SOURCES_OF = {     # kernel sources whose change invalidates a workload's PMC summary
    "ga": ("ga_common.h", "ga_forward_kernel.h", "ga_forward_kernel_v2.h"),
    "ga3": ("ga_common.h", "ga_forward_kernel_v2.h", "ga_forward_kernel_v3.h"),
    "transmil": ("transmil.hip", "transmil_attn.hip", "transmil_pinv.hip", "linear_kernel.h", "linear.hip", "gemm_f32.hip", "gemm_internal.h"),
    "wide": ("ga_common.h", "linear_kernel.h", "linear.hip", "ga_forward_kernel_v2.h", "ga_train.hip"),
    "train": ("ga_common.h", "ga_forward_kernel.h", "ga_forward_kernel_v2.h", "ga_step.hip", "ga_train.hip", "ga_bwd_tile.hip", "ga_backward.hip",
              "ga_pack.hip", "wgrad.hip", "optim.hip", "optim_kernel.h", "ga_opt_step.hip"),
}


def kernel_source_id(workload="ga_eval"):
    """Fingerprint of a workload's kernel sources; tools/pmc_ga.py stamps it into every PMC summary it writes."""
    import hashlib
    h = hashlib.sha1()
    for f in SOURCES_OF.get("wide" if workload == "ga_gigapath" else "ga3" if workload in ("ga_uni", "ga_clip_l") else workload, SOURCES_OF["ga"]):
        with open(os.path.join(ROOT, "acmil_amd", "csrc", f), "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:12]


def pmc_traffic(workload, precision, batch):
    """HBM bytes per launch of the dominant kernel from the committed rocprofv3 PMC summary of this same command
    (profiles/r*_pmc_<workload>_<precision>_b<B>.json, produced by tools/pmc_ga.py: FETCH_SIZE / WRITE_SIZE in their own
    passes, corrected by a known-byte calibration run).  PMC cannot be collected from inside the timed process, so the
    figure is the recorded one -- and only if the summary was taken on THIS kernel source (fingerprint match); otherwise
    None plus the reason."""
    import glob
    pats = ["r*_pmc_%s_%s_b%d.json" % (workload, precision, batch)]
    if workload == "ga_eval":
        pats.append("r*_pmc_bench_%s_b%d.json" % (precision, batch))      # round-1 naming
    cands = []
    for pat in pats:
        cands += glob.glob(os.path.join(ROOT, "profiles", pat))
    for f in sorted(cands, key=os.path.basename, reverse=True):
        try:
            with open(f) as fh:
                js = json.load(fh)
            if js.get("kernel_source_id") != kernel_source_id(workload):
                return None, "PMC summary %s was taken on a different kernel source (%s)" % (os.path.basename(f), js.get("kernel_source_id"))
            return int(js["traffic_bytes_per_launch"]), os.path.basename(f)
        except Exception:
            continue
    return None, "no PMC summary for this workload / precision / batch under profiles/"


def mfma_probe_tflops(dev, iters=1000, launches=40):
    """acmil_mfma_probe: the rate of v_mfma_f32_32x32x16_f16 ALONE (two 4-wave workgroups per CU, pseudo-random operands, no
    memory traffic) on this box, now, in TFLOP/s -- events on the launch stream, back-to-back launches (~13 ms in all, enough for
    the socket to settle at its power cap).  The ceiling of every split-f16 kernel here; reported beside the nominal 2.5 PF."""
    import ctypes
    from acmil_amd import _lib
    lib = _lib.load()
    wgs = 2 * torch.cuda.get_device_properties(dev).multi_processor_count
    sink = torch.empty(wgs * 256, dtype=torch.float32, device=dev)
    n = ctypes.c_longlong(0)
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(5):
        _lib.check(lib.acmil_mfma_probe(iters, wgs, sink.data_ptr(), ctypes.byref(n), st), "acmil_mfma_probe")
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for _ in range(launches):
        lib.acmil_mfma_probe(iters, wgs, sink.data_ptr(), None, st)
    e1.record()
    torch.cuda.synchronize()
    return n.value * 32768.0 * launches / (e0.elapsed_time(e1) * 1e-3) / 1e12


def algorithmic_work(n, d, di, k, c, da=D_ATTN, s_in=4):
    """bytes and flops of one GA eval forward."""
    p = d * di + 2 * (di * da + da) + da * k + k + (k + 1) * (di * c + c)
    nbytes = n * d * s_in + k * n * 4 + 4 * p
    flops = 2 * n * (d * di + 2 * di * da + da * k + k * di)
    return nbytes, flops


def host_cores():
    """(physical cores, logical CPUs) of this host: distinct (physical id, core id) pairs of /proc/cpuinfo; (None, n) if unreadable."""
    logical = os.cpu_count() or 1
    try:
        seen, phys, core = set(), None, None
        with open("/proc/cpuinfo") as fh:
            for line in fh:
                if line.startswith("physical id"):
                    phys = line.split(":")[1].strip()
                elif line.startswith("core id"):
                    core = line.split(":")[1].strip()
                elif not line.strip():
                    if core is not None:
                        seen.add((phys, core))
                    phys = core = None
        if core is not None:
            seen.add((phys, core))
        return (len(seen) or None), logical
    except OSError:
        return None, logical


def _free_port():
    import socket
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


def self_launch_cmd(gpus, argv, port=None):
    """The command `python bench.py --gpus N ...` re-executes itself as when it was NOT started by a launcher:
    one rank per GPU of this node under torch.distributed.run, rendezvous on 127.0.0.1."""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
            "--master-addr", "127.0.0.1", "--master-port", str(port or _free_port()), os.path.abspath(__file__)] + list(argv)


def maybe_self_launch(args, argv):
    """--gpus N > 1 without a launcher environment: spawn the N ranks ourselves and exit with their status."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    import subprocess
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")     # dmabuf IPC: RCCL needs it on this driver
    env.setdefault("OMP_NUM_THREADS", "8")
    raise SystemExit(subprocess.call(self_launch_cmd(args.gpus, argv), env=env))


def _dist_setup(args):
    """One process per GPU.  Returns (world, rank, device).  --dry-run: CPU + gloo (exercises launcher, rendezvous,
    barrier and the max-over-ranks reduction in a container without GPUs; no compute, no JSON `value`)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("WORLD_SIZE=%d does not match --gpus %d" % (world, args.gpus))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if args.dry_run:
        dev = torch.device("cpu")
        if world > 1:
            import torch.distributed as dist
            dist.init_process_group("gloo")
        return world, rank, dev
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=dev)
    return world, rank, dev


def _sync(world, dev):
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
    if dev.type == "cuda":
        torch.cuda.synchronize()


def dry_run(args):
    """Launcher / rendezvous check without a GPU: K empty steps bracketed exactly like the real loop."""
    world, rank, dev = _dist_setup(args)
    dt = _timed(lambda i: None, args, world, dev)
    extra = {}
    if args.workload == "train":
        # the data-parallel fields of the train line, through the same objects (GradBucket.allreduce_mean on gloo, one 833 216-byte
        # bucket); the direct reduction needs GPUs: its keys are present, its check says so
        from acmil_amd import train as T
        p = torch.nn.Parameter(torch.zeros(208303))
        bucket = T.GradBucket([p])
        t0 = time.perf_counter()
        for _ in range(5):
            bucket.allreduce_mean(world)
        extra = {"allreduce_us": round((time.perf_counter() - t0) / 5 * 1e6, 1) if world > 1 else None,
                 "allreduce_bytes": int(bucket.flat.numel() * 4),
                 "direct_reduce": {"ms_per_step": None, "value": None, "first_step_check": "not run (dry run: no GPU)", "slot_memory": None}}
    if rank == 0:
        print(json.dumps(dict({"dry_run": True, "metric": "launcher check (no compute)", "value": None, "n_gpus": world,
                               "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / max(1, args.steps) * 1e3, 6),
                               "workload": args.workload}, **extra)))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


def _timed(step, args, world, dev):
    """W untimed + exactly K timed steps, barrier + synchronize on both sides, max over ranks.  Python's cyclic collector is run
    before and held off during the region: a full collection of this process takes tens of ms, i.e. as long as 20 steps of the
    headline or 200 training steps, and whether one lands inside the region is chance (seen: 2.1 k instead of 11 k slides/s on the
    per-slide module loop).  Nothing in the steps creates reference cycles that would need it."""
    import gc
    for i in range(args.warmup):
        step(i)
    gc.collect()
    gc.disable()
    try:
        _sync(world, dev)
        t0 = time.perf_counter()
        for i in range(args.steps):
            step(i)
        _sync(world, dev)
        dt = time.perf_counter() - t0
    finally:
        gc.enable()
    if world > 1:
        import torch.distributed as dist
        t = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = float(t.item())
    return dt


def other_workloads(args, ctx):
    """Bench lines of the TransMIL eval forward (configs[3]) and the ACMIL training step (configs[4]), same JSON contract.
    Returns the result dict (rank 0 prints it, or nests it under "secondary" of the default line)."""
    world, rank, dev = ctx
    from acmil_amd import _lib
    _lib.check(_lib.load().acmil_check_device(), "acmil_check_device")
    if args.workload == "transmil":
        from acmil_amd import synthetic as S
        from acmil_amd.architecture.transMIL import TransMIL
        N, D, Di, C = 100000, 768, 384, 2

        class Conf:
            D_feat, D_inner, n_class = D, Di, C
        sd = S.transmil_state_dict(D, Di, C, seed=1)
        model = TransMIL(Conf)
        model.load_state_dict(sd)
        model = model.to(dev).eval()
        bags = [torch.randn(1, N, D, generator=torch.Generator().manual_seed(1000 + rank * 4 + i)).to(dev) for i in range(4)]
        with torch.no_grad():
            dt = _timed(lambda i: model(bags[i % 4]), args, world, dev)
            ms_f16v = None
            if not getattr(args, "no_b1", False):      # fp32 bags of fp16 VALUES (the reference's loader): _fc1 drops its W_hi x_lo products
                b16 = [b.half().float() for b in bags]
                a2 = argparse.Namespace(**vars(args)); a2.steps, a2.warmup = max(10, args.steps // 2), 3
                ms_f16v = round(_timed(lambda i: model(b16[i % 4]), a2, world, dev) / a2.steps * 1e3, 3)
                del b16
        side = int(-(-N ** 0.5 // 1)); n = side * side + 1; m = Di // 2; npad = -(-n // m) * m; h, d = 8, Di // 8
        # SURVEY 8(d), re-associated: fc1 + 2 layers x (qkv, 4 head-batched n' x m x d products, out-proj, res-conv, pinv) + PPEG
        flops = 2 * N * D * Di + 2 * (2 * npad * Di * 3 * Di + 4 * (2 * h * npad * m * d) + 2 * npad * Di * Di + 2 * 33 * npad * Di
                                      + 48 * h * m ** 3) + 2 * 83 * side * side * Di
        nbytes = N * D * 4
        t_slide = dt / args.steps
        # The Linear layers and both attention legs execute as split-f16 (3 f16 MFMA products per fp32 product, fp32 accumulate):
        # the pipe they run on is the f16 matrix pipe (2.5 PF dense).  Only the 48 small pinv products stay exact fp32 MFMA.
        g_lin = 2 * N * D * Di + 2 * (2 * npad * Di * 3 * Di + 2 * npad * Di * Di)          # fc1 + per layer qkv, out-proj
        g_att = 2 * (4 * (2 * h * npad * m * d))                                             # the 4 head-batched n' x m x d products
        executed = 3.0 * (g_lin + g_att) + (flops - g_lin - g_att)
        result = {
            "metric": "slides/sec (TransMIL / Nystrom-attention eval forward, N=100000 D=768)", "value": round(world * args.steps / dt, 2),
            "unit": "slides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(t_slide * 1e3, 3),
            "ms_per_step_fp32_bag_of_fp16_values": ms_f16v,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32 (Linear layers and attention legs as split-f16 x3 MFMA products with fp32 accumulate; pinv products exact fp32 MFMA)",
            "data": "synthetic",
            "config": {"workload": "TransMIL eval forward, one slide per step: N=100000 patches, D=768, D_inner=384, 8 heads, 192 landmarks, "
                                   "n_class=2, fp32 bag resident in HBM, 4 bags rotated", "sharding": "independent slides per GPU, no collective"},
            "roofline": {"kernel": "whole forward (gemm_f16x3 / tm_attn1x / tm_attn3x / pinv / stencils)", "bound": "mfma",
                         "achieved": round(flops / t_slide / 1e12, 1), "peak": 2500.0, "unit": "TFLOP/s",
                         "frac": round(flops / t_slide / 1e12 / 2500.0, 4), "traffic": pmc_traffic("transmil", "f16x3", 1)[0],
                         "traffic_source": pmc_traffic("transmil", "f16x3", 1)[1],
                         "executed_tflops": round(executed / t_slide / 1e12, 1),
                         "executed_frac": round(executed / t_slide / 1e12 / 2500.0, 4),
                         "fp32_equivalent_frac": round(flops / t_slide / 1e12 / 157.3, 4),
                         "note": "achieved = algorithmic flops (SURVEY 8d, re-associated: %.1f GFLOP/slide) over the end-to-end forward time; peak = dense "
                                 "f16 MFMA, the pipe the GEMMs and attention legs run on; executed = MFMA flops issued (x3 for the split products); "
                                 "fp32_equivalent_frac = algorithmic flops against the 157.3 TF fp32 MFMA peak (what an exact-fp32 "
                                 "implementation could reach at most); compulsory input %.0f MB" % (flops / 1e9, nbytes / 1e6)},
        }
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            from oracle import transmil_oracle as TO          # the oracle is only ever the CPU baseline / checker
            x = bags[0].cpu()
            t0 = time.perf_counter()
            ref = TO.transmil_forward(x, sd)
            el = time.perf_counter() - t0
            with torch.no_grad():
                err = (model(bags[0]).cpu() - ref["logits"]).abs().max().item()
            result["cpu_baseline"] = {"value": round(1.0 / el, 3), "unit": "slides/s", "cores": torch.get_num_threads(), "kind": "port",
                                      "sample": "1 forward of the same N=100000 bag (%.1f s), torch-CPU oracle" % el}
            result["max_abs_err_vs_oracle"] = err
        return result
    # ---- training step (configs[4]): one bag per rank per step, fused HIP forward/loss/backward, ONE flat-bucket all-reduce, AdamW
    from acmil_amd import synthetic as S
    from acmil_amd import train as T
    N, C = args.train_n, 7
    conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=C, n_token=N_TOKEN, n_masked_patch=10,
                    mask_drop=0.6, arch="ga", precision=args.precision, seed=1, D_feat=D_FEAT, D_inner=D_INNER)
    torch.manual_seed(0)
    model = T.build_model(conf).to(dev).train()
    T.broadcast_parameters(model, world)
    bucket = T.GradBucket(list(model.parameters()))
    opt = T.make_optimizer(model, conf, dev, bucket, lr=conf.lr)          # FlatAdamW: one launch, shares the gradient bucket
    G = max(1, int(getattr(args, "bags_per_step", 1)))
    bags = [S.synthetic_bag(N, D_FEAT, slide_idx=rank * 8 + i)[0].half().to(dev).unsqueeze(0) for i in range(8)]
    labels = [torch.tensor([(rank * 8 + i) % C], device=dev) for i in range(8)]
    groups = []
    if G > 1:
        # --bags-per-step G: G slides per step and rank, ONE mean gradient (acmil_ga_train_step_group: the single-GPU twin of G-rank data
        # parallelism; under DP one all-reduce per G slides).  Resident as the staging ring delivers a training group: rows back to back.
        # 4 rotating groups of distinct bags (> 256 MB Infinity Cache at every size benched)
        for gi in range(4):
            xs = [S.synthetic_bag(N, D_FEAT, slide_idx=rank * 64 + gi * G + j)[0].half() for j in range(G)]
            groups.append(((torch.cat(xs, 0).to(dev), [N] * G), torch.tensor([(rank * 64 + gi * G + j) % C for j in range(G)], device=dev)))
        del xs

    def make_step(bucket, opt):
        def step(i):      # as train.train_one_epoch: range flag left on the device, looked at two steps late (no host read-back per step)
            # one GPU: the step applies AdamW itself -- its closing launch finishes the gradients, updates and re-packs (7 launches per
            # step); data parallel: the bucket all-reduce sits between the gradients and the optimizer's own launch (9 launches + RCCL)
            if G > 1:
                grp, lab = groups[i % len(groups)]
                _, out = model.train_step_batch(grp, lab, guard_flag=opt.guard_flag, optimizer=opt, track_flag=True,
                                                in_step=(world == 1 and not args.train_separate_opt))
            else:
                _, out = model.train_step(bags[i % 8], labels[i % 8], guard_flag=opt.guard_flag,
                                          optimizer=opt, track_flag=True, in_step=(world == 1 and not args.train_separate_opt))
            if out.get("opt_step_id") is None:
                bucket.sync_from_grads()
                bucket.allreduce_mean(world)
                opt.step(track_flag=True)
            if opt.poll_skipped(2):
                raise SystemExit("bench: a synthetic bag left the split-f16 range")
        return step
    if getattr(args, "direct_leg", False):
        # child mode (see below): the same steps with the DIRECT reduction, one JSON line on rank 0, nothing else
        opt.poll_skipped(0)
        os.environ.setdefault("ACMIL_PEER_TIMEOUT_S", "10")     # a benchmark must not sit out the trainer's 120 s patience
        bucket_d = T.GradBucket(list(model.parameters()))
        if bucket_d.enable_direct(rank, world):
            opt_d = T.make_optimizer(model, conf, dev, bucket_d, lr=conf.lr)
            peer = bucket_d.peer
            dt_d = _timed(make_step(bucket_d, opt_d), args, world, dev)      # (its first step is the checked one: compared with all_reduce)
            if opt_d.peer is not None:
                opt_d.peer.check()
            direct = {"ms_per_step": round(dt_d / args.steps * 1e3, 4), "value": round(world * G * args.steps / dt_d, 1), "unit": "slides/s",
                      "first_step_check": peer.verdict, "slot_memory": peer.memory,
                      "note": "gradient reduction fused into the AdamW launch (publish + flag wait + rank-ordered sum over IPC-mapped peer buckets)"
                              if opt_d.peer is not None else "the first-step check failed: these steps ran on torch.distributed all_reduce"}
            peer.close()
        else:
            direct = {"error": "peer mapping unavailable: torch.distributed only"}
        return {"direct_leg": direct}
    dt = _timed(make_step(bucket, opt), args, world, dev)
    # the collective alone (what the step pays for data parallelism): the same flat-bucket all-reduce + mean, back to back, events on
    # the launch stream; None on one GPU (GradBucket.allreduce_mean issues nothing there)
    allreduce_us = None
    if world > 1:
        for _ in range(5):
            bucket.allreduce_mean(world)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        _sync(world, dev)
        e0.record()
        for _ in range(50):
            bucket.allreduce_mean(world)
        e1.record()
        torch.cuda.synchronize()
        t_ar = torch.tensor([e0.elapsed_time(e1) * 1e3 / 50], dtype=torch.float64, device=dev)
        import torch.distributed as dist
        dist.all_reduce(t_ar, op=dist.ReduceOp.MAX)
        allreduce_us = round(float(t_ar.item()), 2)
    # the same steps with the DIRECT reduction (acmil_amd/peer.py: the optimizer launch reads the peers' buckets through IPC-mapped
    # pointers; no collective launch), timed after the RCCL line so that `value` keeps its meaning.  It runs in a CHILD process per rank
    # (own rendezvous port): the path has never crossed a real xGMI link, and a fault in a kernel that reads a wrongly mapped peer
    # buffer must cost this field, not the whole line.  Never fatal, bounded in time.
    direct = None
    if world > 1 and not getattr(args, "direct_reduce", False):
        direct = {"skipped": "opt-in leg: `bench.py --workload train --gpus N --direct-reduce`", "ms_per_step": None, "value": None,
                  "first_step_check": "not run", "slot_memory": None}
    elif world > 1:
        import subprocess
        env = {k: v for k, v in os.environ.items() if not k.startswith("TORCHELASTIC")}      # (no agent store: the child ranks make their own)
        port = int(os.environ.get("MASTER_PORT", "29500"))
        env["MASTER_PORT"] = str(port + 17 if port + 17 < 65000 else port - 17)
        cmd = [sys.executable, os.path.abspath(__file__), "--workload", "train", "--train-n", str(N), "--gpus", str(world), "--steps", str(args.steps),
               "--warmup", str(args.warmup), "--precision", args.precision, "--bags-per-step", str(G), "--direct-leg", "--no-cpu-baseline"]
        try:
            r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=240)
            lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
            if rank == 0:
                direct = json.loads(lines[-1])["direct_leg"] if (r.returncode == 0 and lines) else {
                    "error": "child exited with %d: %s" % (r.returncode, r.stderr.strip().splitlines()[-1][:300] if r.stderr.strip() else "")}
        except subprocess.TimeoutExpired:
            direct = {"error": "child timed out (240 s)"}
        except Exception as e:
            direct = {"error": "%s: %s" % (type(e).__name__, e)}
        _sync(world, dev)
    _, fwd_flops = algorithmic_work(N, D_FEAT, D_INNER, N_TOKEN, C)
    flops = G * fwd_flops * (1.0 + 4.0 / 3.0)       # backward ~ 1.33 x forward (algorithmic, no recompute counted)
    t_step = dt / args.steps
    result = {
        "metric": "slides/sec (ACMIL-ga training step: fwd + STKIM + losses + bwd + grad all-reduce + AdamW, N=%d D=512 C=7)" % N,
        "value": round(world * G * args.steps / dt, 1), "unit": "slides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "bags_per_step": G, "ms_per_slide": round(t_step * 1e3 / G, 4),
        "ms_per_step": round(t_step * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "allreduce_us": allreduce_us, "allreduce_bytes": int(bucket.flat.numel() * 4), "direct_reduce": direct,
        "dtype": "f32 (split-f16 / split-bf16 x3 MFMA products, fp32 accumulate)" if args.precision == "f16x3" else "f32", "data": "synthetic",
        "config": {"workload": ("ACMIL-ga training, one fp16 bag of N=%d patches per GPU per step" % N if G == 1 else
                                "ACMIL-ga training, %d fp16 bags of N=%d patches per GPU per step (ONE mean gradient per step = what %d data-parallel "
                                "ranks compute; rows of a group resident back to back)" % (G, N, G)) +
                               ", D=512, D_inner=256, n_token=5, n_masked_patch=10, mask_drop=0.6, n_class=7, AdamW", "precision": args.precision,
                   "sharding": "slide-level data parallel: %d bag%s per rank and step, one flat 0.83 MB fp32 gradient all-reduce per step (RCCL)" % (G, "" if G == 1 else "s")},
        "roofline": {"kernel": ("whole step (7 launches: acmil_ga_train_step_adamw, optimizer and weight re-pack in the closing launch)" if G == 1 else
                                "whole step (7 launches for %d slides: acmil_ga_train_step_group, optimizer and weight re-pack in the closing launch)" % G)
                     if world == 1 and not args.train_separate_opt
                     else "whole step (9 launches: acmil_ga_train_step%s + optimizer)" % ("" if G == 1 else "_group"), "bound": "mfma", "achieved": round(flops / t_step / 1e12, 1),
                     "peak": 2500.0 if args.precision == "f16x3" else 157.3, "unit": "TFLOP/s",
                     "frac": round(flops / t_step / 1e12 / (2500.0 if args.precision == "f16x3" else 157.3), 4),
                     # PMC summaries of the step: tools/pmc_ga.py --workload train --whole-step with --batch 1 (N = 10 000) / --batch 50
                     # and --extra "--train-n 50000" (the batch argument only names the file for this workload)
                     # (group steps: --batch 100 + G / 500 + G name the files; traffic is per STEP = G slides)
                     "traffic": pmc_traffic("train", args.precision, ({10000: 1, 50000: 50}.get(N, -1) if G == 1 else {10000: 100 + G, 50000: 500 + G}.get(N, -1)))[0],
                     "traffic_source": pmc_traffic("train", args.precision, ({10000: 1, 50000: 50}.get(N, -1) if G == 1 else {10000: 100 + G, 50000: 500 + G}.get(N, -1)))[1],
                     "algorithmic_bytes": int(G * (2 * N * D_FEAT * 2 + N_TOKEN * N * 4)),
                     "note": "algorithmic flops = 2.33 x forward (SURVEY 8d) over the end-to-end step time"},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ga_oracle as O                     # the oracle is only ever the CPU baseline / checker
        sd = {k: v.detach().cpu().clone().requires_grad_(True) for k, v in model.state_dict().items()}
        xs = [b.float().cpu() for b in bags[:2]]
        torch.set_num_threads(min(16, torch.get_num_threads()))

        def cpu_step(i):
            out = O.acmil_ga_forward(xs[i % 2], sd, n_token=N_TOKEN, n_masked_patch=10, mask_drop=0.6,
                                     uniforms=torch.rand(N_TOKEN, 10), training=True)
            l0, l1, dl = O.acmil_losses(out["sub_preds"], out["slide_pred"], out["A_out"], labels[i % 2].cpu(), N_TOKEN)
            for v in sd.values():
                v.grad = None
            (l0 + l1 + dl).backward()
        try:
            cpu_step(0)
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 10.0:
                cpu_step(n); n += 1
            el = time.perf_counter() - t0
            result["cpu_baseline"] = {"value": round(n / el, 2), "unit": "slides/s", "cores": torch.get_num_threads(), "kind": "port",
                                      "sample": "%d forward+backward steps on the same bags (%.1f s), torch-CPU oracle + autograd, no optimizer" % (n, el)}
        except TypeError as e:   # oracle signature drift must not kill the bench line
            result["cpu_baseline"] = {"value": None, "unit": "slides/s", "cores": 0, "kind": "port", "sample": "unavailable: %s" % e}
    return result


WIDE_SHAPES = {   # the reference's widest feature extractor (Step3_WSI_classification_ACMIL.py:78-87): no single-kernel family, composed path
    "ga_gigapath": dict(N=50000, D=1536, Di=768, K=5, C=2, name="GigaPath (ViT-g, 1536 -> 768)"),
}


def wide_workload(args, ctx):
    """ACMIL-ga eval forward at the wide D_inner families, one slide per step through the product module (`model(x)`: packed-weight
    projection kernel -> gated scores -> pooling -> merge + heads; h [N, D_inner] makes one HBM round trip).  Same JSON contract."""
    world, rank, dev = ctx
    from acmil_amd import _lib, ops
    from acmil_amd import synthetic as S
    from acmil_amd.architecture.transformer import ACMIL_GA
    _lib.check(_lib.load().acmil_check_device(), "acmil_check_device")
    sh = WIDE_SHAPES[args.workload]
    N, D, Di, K, C = sh["N"], sh["D"], sh["Di"], sh["K"], sh["C"]

    class _Conf:
        D_feat, D_inner, n_class, n_token = D, Di, C, K
    sd_cpu = S.ga_state_dict(D, Di, C, K, seed=0)
    model = ACMIL_GA(_Conf, n_token=K, n_masked_patch=10, mask_drop=0.6, precision=args.precision)
    model.load_state_dict(sd_cpu)
    model = model.to(dev).eval()
    nb = 8
    bags = [S.synthetic_bag(N, D, slide_idx=rank * nb + i)[0].to(dev) for i in range(nb)]       # 8 x 205 / 307 MB: beyond the 256 MB MALL
    torch.cuda.synchronize()
    with torch.no_grad():
        # the projection kernel alone (events on the launch stream), first: it also brings the device to operating clocks
        # (--no-b1: skipped -- the PMC passes of tools/pmc_ga.py want the step's own launches only)
        t_lin = None
        if not args.no_b1:
            w1 = model._packed_w1()
            hbuf = torch.empty(N, Di, dtype=torch.float32, device=dev)
            for i in range(5):
                ops.linear_f16x3(bags[i % nb], w1, Di, relu=True, out=hbuf)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            n_k = 40
            for i in range(n_k):
                ops.linear_f16x3(bags[i % nb], w1, Di, relu=True, out=hbuf)
            e1.record()
            torch.cuda.synchronize()
            t_lin = e0.elapsed_time(e1) * 1e-3 / n_k
        # --batch G > 1 (default: groups of 16): a step is ONE group of G slides whose rows lie back to back, as staging.staged_train_groups
        # delivers them to train.evaluate (ACMIL_GA.forward_group: one projection launch + one gated-score launch over all rows, pooling
        # tiles per bag, merge + heads of all bags); --batch 1: the reference's `model(x)` per slide.  3 rotating groups of 16 x 307 MB.
        G = max(1, min(ops.MAX_GROUP, int(args.batch)))
        per_slide = None
        if G > 1:
            groups = []
            for gi in range(3):      # filled row range by row range with device-to-device copies (no concatenation kernel in a PMC pass)
                xg = torch.empty(G * N, D, dtype=bags[0].dtype, device=dev)
                for j in range(G):
                    xg[j * N:(j + 1) * N].copy_(bags[(gi * 3 + j) % nb])
                groups.append((xg, [N] * G))
            torch.cuda.synchronize()
            dt = _timed(lambda i: model.forward_group(*groups[i % 3]), args, world, dev)
            if not args.no_b1:
                a1 = argparse.Namespace(**vars(args)); a1.steps, a1.warmup = 50, 5
                per_slide = round(world * a1.steps / _timed(lambda i: model(bags[i % nb].unsqueeze(0)), a1, world, dev), 1)
        else:
            dt = _timed(lambda i: model(bags[i % nb].unsqueeze(0)), args, world, dev)
        # what the reference's loop feeds (Step3_WSI_classification_ACMIL.py:193: fp16-stored features up-cast to fp32): the same bags
        # rounded to fp16 values, still fp32 storage -- the Linear kernel drops the W_hi x_lo products of such rows per wave and K step
        rate_f16v = rate_f16s = None
        if not args.no_b1:
            a2 = argparse.Namespace(**vars(args)); a2.steps, a2.warmup = max(10, args.steps // 2), 3
            if G > 1:
                for xg, _ in groups:
                    for j in range(G):
                        xg[j * N:(j + 1) * N].copy_(xg[j * N:(j + 1) * N].half())
                rate_f16v = round(world * G * a2.steps / _timed(lambda i: model.forward_group(*groups[i % 3]), a2, world, dev), 1)
                # ... and as the staging ring delivers them when the features are stored fp16: no up-cast at all (train.evaluate)
                groups = [(xg.half(), r) for xg, r in groups]
                rate_f16s = round(world * G * a2.steps / _timed(lambda i: model.forward_group(*groups[i % 3]), a2, world, dev), 1)
            else:
                b16 = [b.half().float() for b in bags]
                rate_f16v = round(world * a2.steps / _timed(lambda i: model(b16[i % nb].unsqueeze(0)), a2, world, dev), 1)
                del b16
    t_slide = dt / args.steps / G
    nbytes, flops = algorithmic_work(N, D, Di, K, C)
    g1 = 2.0 * N * D * Di
    executed = 3.0 * flops if args.precision == "f16x3" else flops
    peak = 2500.0 if args.precision == "f16x3" else 157.3
    result = {
        "metric": "slides/sec (ACMIL-ga eval forward, N=%d D=%d D_inner=%d: %s)" % (N, D, Di, sh["name"]),
        "value": round(world * G * args.steps / dt, 1), "unit": "slides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(t_slide * G * 1e3, 4), "ms_per_slide": round(t_slide * 1e3, 4), "slides_per_step": G,
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32 (projections as split-f16 x3 MFMA products, fp32 accumulate)" if args.precision == "f16x3" else "f32", "data": "synthetic",
        "config": {"workload": ("ACMIL-ga eval forward, one slide per step through the module" if G == 1 else
                                "ACMIL-ga eval forward, %d slides per step in grouped launches (rows of a group resident back to back, as "
                                "train.evaluate stages them; ACMIL_GA.forward_group)" % G) +
                               ": N=%d patches, D=%d, D_inner=%d, n_token=%d, n_class=%d, fp32 bags resident in HBM, %s; composed kernels "
                               "(projection -> gated scores -> pooling)" % (N, D, Di, K, C, "%d bags rotated" % nb if G == 1 else "3 groups rotated"),
                   "precision": args.precision, "slides_per_step": G, "sharding": "independent slides per GPU, no collective"},
        "module_slides_per_s": None if per_slide is None else {"model(x) per slide": per_slide},
        "slides_per_s_fp32_bags_of_fp16_values": rate_f16v, "slides_per_s_fp16_stored_bags": rate_f16s,
        "roofline": {"kernel": "whole composed forward (lin_kernel projection + gated scores + pooling + merge + heads)", "bound": "mfma",
                     "achieved": round(flops / t_slide / 1e12, 2), "peak": peak, "unit": "TFLOP/s", "frac": round(flops / t_slide / 1e12 / peak, 4),
                     # PMC summary of the whole composed forward: tools/pmc_ga.py --workload <name> --whole-step --batch 1
                     # (per STEP: G slides; tools/pmc_ga.py --workload <name> --whole-step --batch G)
                     "traffic": pmc_traffic(args.workload, args.precision, G)[0], "traffic_source": pmc_traffic(args.workload, args.precision, G)[1],
                     "algorithmic_bytes": int(G * nbytes),
                     "executed_tflops": round(executed / t_slide / 1e12, 1), "executed_frac": round(executed / t_slide / 1e12 / peak, 4),
                     "projection_kernel": None if t_lin is None else {
                         "us_per_launch": round(t_lin * 1e6, 1), "achieved_tflops": round(g1 / t_lin / 1e12, 1),
                         "executed_frac": round((3.0 if args.precision == "f16x3" else 1.0) * g1 / t_lin / 1e12 / peak, 4),
                         "hbm_gbs": round((N * D * 4 + N * Di * 4) / t_lin / 1e9, 1)},
                     "hbm": {"achieved": round((nbytes + 2.0 * N * Di * 4) / t_slide / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                             "frac": round((nbytes + 2.0 * N * Di * 4) / t_slide / 1e9 / 8000.0, 4),
                             "note": "algorithmic bytes + the h [N, D_inner] round trip of the composed path"},
                     "note": "flops = algorithmic (SURVEY 8d: %.2f GFLOP/slide); executed = x3 for the split products" % (flops / 1e9)},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ga_oracle as O                     # the oracle is only ever the CPU baseline / checker
        torch.set_num_threads(min(16, torch.get_num_threads()))
        x0 = bags[0].cpu().unsqueeze(0)
        with torch.no_grad():
            O.acmil_ga_forward(x0, sd_cpu, n_token=K)
            t0 = time.perf_counter(); n = 0
            while time.perf_counter() - t0 < 8.0:
                ref = O.acmil_ga_forward(x0, sd_cpu, n_token=K); n += 1
            el = time.perf_counter() - t0
            got = model(bags[0].unsqueeze(0))
        result["cpu_baseline"] = {"value": round(n / el, 2), "unit": "slides/s", "cores": torch.get_num_threads(), "kind": "port",
                                  "sample": "%d forwards of one N=%d D=%d bag (%.1f s), torch-CPU oracle" % (n, N, D, el)}
        result["max_abs_err_vs_oracle"] = max((got[2].cpu() - ref["A_out"]).abs().max().item(), (got[0].cpu() - ref["sub_preds"]).abs().max().item())
    return result


# GA eval-forward workloads: the BASELINE.json headline and configs[2]
GA_SHAPES = {
    "ga_eval": dict(N=50000, D=512, Di=256, K=5, C=2, xdtype="float32",
                    metric="slides/sec (ACMIL-ga attention-aggregation forward, N=50000 D=512)"),
    "ga_cfg3": dict(N=50000, D=384, Di=128, K=5, C=2, xdtype="bfloat16",
                    metric="slides/sec (ACMIL-ga forward, Camelyon16-shape bags N=50000 D=384 D_inner=128, bf16 bags)"),
    # the wide families with a fused kernel since round 5 (csrc/ga_forward_kernel_v3.h: one 512-register wave per SIMD); the step is the
    # headline's: one fused launch of --batch bags (as acmil_amd.train.evaluate drives the module); `model(x)` per slide rides along
    "ga_uni": dict(N=50000, D=1024, Di=512, K=5, C=2, xdtype="float32",
                   metric="slides/sec (ACMIL-ga eval forward, N=50000 D=1024 D_inner=512: UNI (ViT-L/16, 1024 -> 512))"),
    "ga_clip_l": dict(N=50000, D=768, Di=384, K=5, C=2, xdtype="float32",
                      metric="slides/sec (ACMIL-ga eval forward, N=50000 D=768 D_inner=384: CLIP-L-336 (768 -> 384))"),
}


def main(argv=None):
    argv = list(sys.argv[1:] if argv is None else argv)
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--precision", default="f16x3", choices=["f16x3", "fp32", "f16"],
                    help="arithmetic of the two projection GEMMs; f16x3 and fp32 are inside the 1e-4 fp32 parity bound, "
                         "f16 (single pass, ~2e-4 on the scores) is a throughput mode and is labelled as such")
    ap.add_argument("--batch", type=int, default=64,
                    help="slides per step (1..64): bags of one step go through ONE fused launch (acmil_ga_forward_batch); 1 = the "
                         "reference's strictly per-slide call pattern.  A launch of 64 bags is 48 rounds of tiles on the 512 "
                         "persistent workgroups: the ragged last round and merge + heads weigh 2 %% instead of 9 %% at 16 bags")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--workload", default="ga_eval", choices=["ga_eval", "ga_cfg3", "transmil", "train", "ga_uni", "ga_gigapath", "ga_clip_l"],
                    help="ga_eval = the BASELINE.json headline (default); ga_cfg3 = configs[2] (N=50000, D=384, D_inner=128, bf16 "
                         "bags); transmil = configs[3] (N=100000, D=768 TransMIL eval forward); train = configs[4] (ACMIL "
                         "training step, slide-level DP, gradient all-reduce)")
    ap.add_argument("--train-n", type=int, default=10000, help="patches per bag of the train workload")
    ap.add_argument("--no-b1", action="store_true",
                    help="skip the one-slide-per-call latency loop (profiling runs: keeps a single grid shape per kernel name)")
    ap.add_argument("--no-secondary", action="store_true",
                    help="default workload only: skip the nested `secondary` lines (configs[2], [3], [4])")
    ap.add_argument("--bags-per-step", type=int, default=1,
                    help="train workload: slides per step and rank, ONE mean gradient per step (acmil_ga_train_step_group); 1 = the reference's B = 1 SGD")
    ap.add_argument("--train-separate-opt", action="store_true",
                    help="train workload on one GPU: keep the optimizer as its own launch (the data-parallel launch sequence) -- A/B of the in-step optimizer")
    ap.add_argument("--direct-leg", action="store_true", help=argparse.SUPPRESS)      # child mode of the train workload (direct gradient reduction)
    ap.add_argument("--direct-reduce", action="store_true",
                    help="train workload at --gpus N > 1: ALSO time the steps with the direct (IPC-mapped peer bucket) gradient reduction, in a "
                         "child process per rank.  Opt-in: the path has never crossed a real xGMI link, and the driver's scaling run of the "
                         "default line must not depend on it (`direct_reduce` is then {\"skipped\": ...})")
    ap.add_argument("--dry-run", action="store_true",
                    help="launcher / rendezvous check on CPU with gloo: no GPU, no compute, no metric value")
    args = ap.parse_args(argv)
    maybe_self_launch(args, argv)         # --gpus N > 1 and no launcher: re-execute under torch.distributed.run
    if args.dry_run:
        return dry_run(args)
    ctx = _dist_setup(args)
    world, rank, dev = ctx
    if args.workload in ("transmil", "train"):
        result = other_workloads(args, ctx)
    elif args.workload in WIDE_SHAPES:
        result = wide_workload(args, ctx)
    else:
        result = ga_workload(args, ctx)
        if args.workload == "ga_eval" and not args.no_secondary:
            result["secondary"] = secondary_lines(args, ctx)
    if rank == 0:
        print(json.dumps(result))
    if world > 1:
        import torch.distributed as dist
        dist.destroy_process_group()


SECONDARY = (     # (key, argv overrides): the other BASELINE.json configs, measured AFTER the headline's timed region, on the same ranks
    ("ga_cfg3", dict(workload="ga_cfg3", steps=20, warmup=5)),
    ("transmil", dict(workload="transmil", steps=30, warmup=5)),
    ("train_n10k", dict(workload="train", train_n=10000, steps=300, warmup=50)),
    ("train_n50k", dict(workload="train", train_n=50000, steps=150, warmup=30)),
    ("train_n10k_g8", dict(workload="train", train_n=10000, steps=100, warmup=20, bags_per_step=8)),
    ("train_n50k_g8", dict(workload="train", train_n=50000, steps=40, warmup=8, bags_per_step=8)),
    # the reference's widest extractor (Step3_WSI_classification_ACMIL.py:84-87), the one family without a fused kernel: groups of 16 slides
    ("ga_gigapath_g16", dict(workload="ga_gigapath", steps=20, warmup=3, batch=16)),
)


def secondary_lines(args, ctx):
    """configs[2], [3], [4] on the driver-run line: each entry is the full JSON line `bench.py --workload <w>` would print with the
    same --gpus (own `value`, `ms_per_step`, `roofline` incl. `traffic`), without the CPU-baseline and latency side legs.  They run
    after the headline has been measured, so `value` / `ms_per_step` of the line itself are untouched.  At --gpus N > 1 every rank
    takes part (the training step then includes its gradient all-reduce: `allreduce_us`)."""
    import copy
    import gc
    out = {}
    for key, over in SECONDARY:
        a = copy.copy(args)
        a.no_cpu_baseline = True
        a.no_b1 = True
        a.direct_reduce = False       # never on the driver's default line (see --direct-reduce)
        a.batch = 64
        a.precision = "f16x3"
        for k, v in over.items():
            setattr(a, k, v)
        t0 = time.perf_counter()
        try:
            r = ga_workload(a, ctx) if a.workload in GA_SHAPES else wide_workload(a, ctx) if a.workload in WIDE_SHAPES else other_workloads(a, ctx)
            r["wall_s"] = round(time.perf_counter() - t0, 1)
        except (Exception, SystemExit) as e:      # a secondary line must never cost the headline
            r = {"error": "%s: %s" % (type(e).__name__, e)}
        out[key] = r
        gc.collect()
        if ctx[2].type == "cuda":
            torch.cuda.empty_cache()
    return out


def ga_workload(args, ctx):
    """The GA eval-forward line (BASELINE.json headline / configs[2]); returns the result dict."""
    world, rank, dev = ctx
    if world > 1:
        import torch.distributed as dist

    from acmil_amd import _lib, ops
    from acmil_amd import synthetic as S

    shape = GA_SHAPES[args.workload]
    N_PATCH, D_FEAT, D_INNER, N_TOKEN, N_CLASS = shape["N"], shape["D"], shape["Di"], shape["K"], shape["C"]
    x_dtype = getattr(torch, shape["xdtype"])
    s_in = 4 if x_dtype == torch.float32 else 2
    _lib.check(_lib.load().acmil_check_device(), "acmil_check_device")
    sd_cpu = S.ga_state_dict(D_FEAT, D_INNER, N_CLASS, N_TOKEN, seed=0)
    sd = {k: v.to(dev) for k, v in sd_cpu.items()}
    packed, dims = ops.ga_pack_weights(
        sd["dimreduction.fc1.weight"], sd["attention.attention_V.0.weight"], sd["attention.attention_V.0.bias"],
        sd["attention.attention_U.0.weight"], sd["attention.attention_U.0.bias"],
        sd["attention.attention_weights.weight"], sd["attention.attention_weights.bias"],
        [sd["classifier.%d.fc.weight" % i] for i in range(N_TOKEN)],
        [sd["classifier.%d.fc.bias" % i] for i in range(N_TOKEN)],
        sd["Slide_classifier.fc.weight"], sd["Slide_classifier.fc.bias"], args.precision)
    # resident synthetic bags: slide index = rank * N_BAGS + i (disjoint across ranks)
    B = max(1, min(64, args.batch))
    N_BAGS = max(16, B)          # every bag of a launch is a distinct resident bag (64 x 102 MB = 6.5 GB of the 288 GB)
    bags = [S.synthetic_bag(N_PATCH, D_FEAT, slide_idx=rank * N_BAGS + i)[0].to(x_dtype).to(dev) for i in range(N_BAGS)]
    torch.cuda.synchronize()

    def step(i):
        if B == 1:
            return ops.ga_forward(bags[i % N_BAGS], packed, dims, args.precision)
        return ops.ga_forward_batch([bags[(i * B + j) % N_BAGS] for j in range(B)], packed, dims, args.precision)

    # ---- dominant kernel alone (ga_fwd_kernel, same template instance): scores-only calls launch just it
    # (measured FIRST: ~50 back-to-back launches also bring the GPU from its idle power state to operating clocks, so the W warm-up +
    #  K timed steps below see a device in steady state -- with the driver's --steps 20 --warmup 5 the whole timed region is
    #  ~20 ms, and on a cold device the DPM ramp alone cost ~10 % of it: 16.6 k vs 18.3 k slides/s on the same box)
    n_k = max(50, min(args.steps, 400))
    ws = torch.zeros(_lib.load().acmil_ga_workspace_bytes(N_PATCH, D_FEAT, D_INNER, N_TOKEN, N_CLASS, ops.mode_id(args.precision)),
                     dtype=torch.uint8, device=dev)      # zeroed once: the control block contract of the GA workspace
    a_out = torch.empty(N_TOKEN, N_PATCH, dtype=torch.float32, device=dev)
    stream = torch.cuda.current_stream().cuda_stream

    import ctypes
    ws_b = torch.zeros(_lib.load().acmil_ga_batch_workspace_bytes(B, (ctypes.c_int * B)(*[N_PATCH] * B), D_FEAT, D_INNER, N_TOKEN,
                                                                  N_CLASS, ops.mode_id(args.precision)), dtype=torch.uint8, device=dev)
    a_outs = [torch.empty(N_TOKEN, N_PATCH, dtype=torch.float32, device=dev) for _ in range(B)]
    a_ptrs = (ctypes.c_void_p * B)(*[t.data_ptr() for t in a_outs])
    ns_arr = (ctypes.c_int * B)(*[N_PATCH] * B)

    def main_kernel(i):
        # same launch as the timed steps (same template instance, same grid: B bags), without merge / heads
        xp = (ctypes.c_void_p * B)(*[bags[(i * B + j) % N_BAGS].data_ptr() for j in range(B)])
        rc = _lib.load().acmil_ga_forward_batch(B, xp, ns_arr, ops._DT[x_dtype], packed.data_ptr(), *dims.args(),
                                                ops.mode_id(args.precision), a_ptrs, None, None, None, None, 1,
                                                ws_b.data_ptr(), stream)
        _lib.check(rc, "acmil_ga_forward_batch")

    for i in range(10):
        main_kernel(i)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    e0.record()
    for i in range(n_k):
        main_kernel(i)
    e1.record()
    torch.cuda.synchronize()
    t_kernel = e0.elapsed_time(e1) * 1e-3 / n_k  # seconds per launch (back-to-back launches on the launch stream)
    # the matrix pipe alone on this box, right after the kernel (same power state): the ceiling `executed` is to be read against
    probe_tf = mfma_probe_tflops(dev) if args.precision != "fp32" else None


    dt = _timed(step, args, world, dev)   # W untimed + exactly K timed steps, barrier + synchronize both sides, max over ranks
    slides_per_s = world * args.steps * B / dt

    # per-slide latency in the reference's B=1 call pattern (one slide per call, calls back to back)
    ms_b1 = None
    if not args.no_b1:
        n_lat = 100
        for i in range(10):
            ops.ga_forward(bags[i % N_BAGS], packed, dims, args.precision)
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(n_lat):
            ops.ga_forward(bags[i % N_BAGS], packed, dims, args.precision)
        torch.cuda.synchronize()
        ms_b1 = (time.perf_counter() - t1) / n_lat * 1e3

    # ---- the PRODUCT's own eval path (what a user of the drop-in module gets): `model(x)` per slide exactly as the reference's
    # evaluate loop calls it (Step3_WSI_classification_ACMIL.py:253-268; device-side range guard, no host read-back), and
    # `model.forward_batch` as acmil_amd.train.evaluate drives it (EVAL_BATCH bags per launch, range word looked at one batch late)
    module_rates = None
    if not args.no_b1 and world == 1:
        from acmil_amd.architecture.transformer import ACMIL_GA

        class _Conf:
            D_feat, D_inner, n_class, n_token = D_FEAT, D_INNER, N_CLASS, N_TOKEN
        model = ACMIL_GA(_Conf, n_token=N_TOKEN, n_masked_patch=10, mask_drop=0.6, precision=args.precision)
        model.load_state_dict(sd_cpu)
        model = model.to(dev).eval()
        import gc
        gc.collect()
        gc.disable()      # a full collection of this process (torch + modules: ~40 ms) inside the 100-call loop read as 2.1 k instead of 11 k slides/s
        with torch.no_grad():
            for i in range(5):
                model(bags[i % N_BAGS].unsqueeze(0))
            torch.cuda.synchronize()
            tm = time.perf_counter()
            n_m1 = 100
            for i in range(n_m1):
                model(bags[i % N_BAGS].unsqueeze(0))
            torch.cuda.synchronize()
            rate_b1 = n_m1 / (time.perf_counter() - tm)
            pend = None
            from acmil_amd.train import EVAL_BATCH as EB
            for i in range(3):      # warm-up as the loop runs (deferred guard: the pinned status words exist before the timed region)
                _, pend = model.forward_batch([bags[(i * EB + j) % N_BAGS] for j in range(EB)], defer_guard=True)
                int(pend)
            pend = None
            torch.cuda.synchronize()
            tm = time.perf_counter()
            n_mb = max(10, args.steps)
            for i in range(n_mb):
                _, status = model.forward_batch([bags[(i * EB + j) % N_BAGS] for j in range(EB)], defer_guard=True)
                if pend is not None and int(pend) != 0:
                    raise SystemExit("bench: a synthetic bag left the split-f16 range")
                pend = status
            torch.cuda.synchronize()
            rate_b16 = n_mb * EB / (time.perf_counter() - tm)
        gc.enable()
        module_rates = {"model(x) per slide": round(rate_b1, 1), "model.forward_batch x%d (as train.evaluate)" % EB: round(rate_b16, 1),
                        "range_fallbacks": int(model.range_fallbacks)}      # (device-side repeats that actually ran: 0 on in-range bags)
        del model

    # ---- data-faithful variant: the same bags as they are stored on disk, fp16 (Step2_feature_extract.py:165);
    # the kernel converts in registers and the x_lo product vanishes.  Reported next to the fp32-bag headline, never as `value`.
    sps_fp16 = None
    if not args.no_b1 and world == 1 and x_dtype == torch.float32:
        bags16 = [b.half() for b in bags]
        step16 = lambda i: ops.ga_forward_batch([bags16[(i * B + j) % N_BAGS] for j in range(B)], packed, dims, args.precision) \
            if B > 1 else ops.ga_forward(bags16[i % N_BAGS], packed, dims, args.precision)
        for i in range(5):
            step16(i)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n16 = max(20, args.steps // 2)
        for i in range(n16):
            step16(i)
        torch.cuda.synchronize()
        sps_fp16 = n16 * B / (time.perf_counter() - t2)
        del bags16
    # ---- ... and as the reference's LOOP hands them to the module: fp16-stored values up-cast to fp32 (Step3_WSI_classification_ACMIL.py:193
    # `.to(device, dtype=torch.float32)`): fp32 bytes, but every x_lo half is an exact zero -- the kernel notices per wave and K step
    # and skips the W_hi x_lo products (same results).  Measured last: it rounds the resident bags in place.
    sps_fp32_exact = None
    if not args.no_b1 and world == 1 and x_dtype == torch.float32 and args.precision == "f16x3":
        for b in bags:
            b.copy_(b.half())
        for i in range(5):
            step(i)
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        n16 = max(20, args.steps // 2)
        for i in range(n16):
            step(i)
        torch.cuda.synchronize()
        sps_fp32_exact = n16 * B / (time.perf_counter() - t2)

    nbytes, flops = algorithmic_work(N_PATCH, D_FEAT, D_INNER, N_TOKEN, N_CLASS, s_in=s_in)
    nbytes, flops = nbytes * B, flops * B            # one launch processes B slides
    split = args.precision == "f16x3"
    mfma_peak = 157.3 if args.precision == "fp32" else 2500.0   # TFLOP/s dense: fp32 MFMA / f16 MFMA
    # MFMA flops actually executed: split-f16 runs 3 products per fp32 product (2 in GEMM1 when the bag is fp16 / bf16: x_lo = 0)
    g1 = 2.0 * N_PATCH * D_FEAT * D_INNER * B
    g2 = 2.0 * N_PATCH * 2 * D_INNER * D_ATTN * B
    p1 = (3.0 if x_dtype == torch.float32 else 2.0) if split else 1.0       # 16-bit bags (fp16 and, round 4, bf16) have no x_lo product
    executed = g1 * p1 + g2 * (3.0 if split else 1.0)
    version = 1 if (os.environ.get("ACMIL_GA_KERNEL") == "1" or not split) else 2
    if version == 2:      # csrc/ga_forward.hip::ga_v2_waves: 4 waves unless the opt-in is set
        waves = 8 if os.environ.get("ACMIL_GA2_WAVES") == "8" else 4
    else:                 # ga_pick_waves of the round-1 kernel
        waves = 4 if (B * N_PATCH >= 1024 * 128 or N_PATCH < 32768) else 8
    traffic, traffic_src = pmc_traffic(args.workload, args.precision, B)
    v3 = split and D_INNER in (384, 512)      # csrc/ga_forward.hip::ga_pick_v3: the wide families run the one-wave-per-SIMD kernel
    roofline = {
        "kernel": ("ga_fwd3_kernel<ND=%d,PB=1,KP=%d,x=%s>, %d bags per launch" % (D_INNER // 32, 5 if N_TOKEN > 1 else 1, shape["xdtype"], B)) if v3 else
                  "%s<ND=%d,KP=%d,%s,x=%s,waves=%d>, %d bags per launch" % (
            "ga_fwd2_kernel" if version == 2 else "ga_fwd_kernel", D_INNER // 32, 5 if N_TOKEN > 1 else 1, args.precision,
            shape["xdtype"], waves, B),
        "bound": "mfma",
        "achieved": round(flops / t_kernel / 1e12, 2), "peak": mfma_peak, "unit": "TFLOP/s",
        "frac": round(flops / t_kernel / 1e12 / mfma_peak, 4),
        "traffic": traffic, "traffic_source": traffic_src,
        "us_per_launch": round(t_kernel * 1e6, 2),
        "executed_tflops": round(executed / t_kernel / 1e12, 1),
        "executed_frac": round(executed / t_kernel / 1e12 / mfma_peak, 4),
        "mfma_only_probe_tflops": None if probe_tf is None else round(probe_tf, 1),
        "mfma_only_probe_frac_of_peak": None if probe_tf is None else round(probe_tf / mfma_peak, 4),
        "executed_frac_of_probe": None if probe_tf is None else round(executed / t_kernel / 1e12 / probe_tf, 4),
        "note": "flops = algorithmic (SURVEY 8d: %.2f GFLOP/slide x slides per launch); executed = MFMA flops issued "
                "(split-f16: 3 f16 products per fp32 product)" % (flops / B / 1e9),
        "hbm": {"achieved": round(nbytes / t_kernel / 1e9, 1), "peak": 8000.0, "unit": "GB/s",
                "frac": round(nbytes / t_kernel / 1e9 / 8000.0, 4), "algorithmic_bytes": nbytes},
    }
    dtype_str = {"f16x3": "f32 (projections as split-f16 x3 MFMA products, fp32 accumulate)", "fp32": "f32",
                 "f16": "f16 (single-pass f16 MFMA, fp32 accumulate: THROUGHPUT MODE, scores ~2e-4 from the fp32 reference)"}[args.precision]
    if x_dtype != torch.float32:
        dtype_str = "%s bags; %s" % (shape["xdtype"], dtype_str)

    result = {
        "metric": shape["metric"],
        "value": round(slides_per_s, 1), "unit": "slides/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": round(dt / args.steps * 1e3, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": dtype_str,
        "data": "synthetic",
        "config": {"workload": "ACMIL-ga eval forward, %d slide(s) per step in one fused launch: N=%d patches, D=%d, "
                               "D_inner=%d, n_token=%d, n_class=%d, %s bags resident in HBM, %d bags rotated" % (
                                   B, N_PATCH, D_FEAT, D_INNER, N_TOKEN, N_CLASS, shape["xdtype"], N_BAGS),
                   "precision": args.precision, "slides_per_step": B, "sharding": "independent slides per GPU, no collective"},
        "attention_fwd_ms_per_slide": round(dt / (args.steps * B) * 1e3, 4),
        "attention_fwd_ms_per_slide_b1": None if ms_b1 is None else round(ms_b1, 4),
        "slides_per_s_fp16_stored_bags": None if sps_fp16 is None else round(sps_fp16, 1),
        "slides_per_s_fp32_bags_of_fp16_values": None if sps_fp32_exact is None else round(sps_fp32_exact, 1),
        "module_slides_per_s": module_rates,
        "roofline": roofline,
    }

    # ---- CPU baseline: the oracle (port of the reference's PyTorch-CPU forward), host cores of this box, rank 0, N=1 only
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import ga_oracle as O                     # the oracle is only ever the CPU baseline / checker
        all_cores = torch.get_num_threads()
        xs = [b.float().cpu().unsqueeze(0) for b in bags[:4]]

        def cpu_run(threads, budget_s, min_iters):
            torch.set_num_threads(threads)
            with torch.no_grad():
                for i in range(2):
                    O.acmil_ga_forward(xs[i % 4], sd_cpu, n_token=N_TOKEN)
                n, t0c = 0, time.perf_counter()
                while True:
                    O.acmil_ga_forward(xs[n % 4], sd_cpu, n_token=N_TOKEN)
                    n += 1
                    e = time.perf_counter() - t0c
                    if (e > budget_s and n >= min_iters) or e > 3 * budget_s:
                        return n, e

        # torch's intra-op threading is not monotone in the thread count at this size: probe a few settings briefly,
        # then time the best one on the bounded sample (fair to the CPU: its best configuration is the baseline)
        probe = {}
        for th in sorted({all_cores, max(1, all_cores // 2), 32, 16, 8} & set(range(1, all_cores + 1))):
            n, e = cpu_run(th, 1.5, 3)
            probe[th] = n / e
        cores = max(probe, key=probe.get)
        n_cpu, el = cpu_run(cores, 10.0, 10)
        torch.set_num_threads(all_cores)
        cpu_sps = n_cpu / el
        # cross-check while we are here: GPU result of the last step vs the oracle on the same bag
        last1 = ops.ga_forward(bags[3], packed, dims, args.precision)
        ref = O.acmil_ga_forward(bags[3].float().cpu().unsqueeze(0), sd_cpu, n_token=N_TOKEN)
        err = max((last1["A_out"].cpu() - ref["A_out"][0]).abs().max().item(),
                  (last1["sub_preds"].cpu() - ref["sub_preds"]).abs().max().item())
        result["cpu_baseline"] = {"value": round(cpu_sps, 2), "unit": "slides/s", "cores": cores, "kind": "port",
                                  "sample": "%d forwards of the same N=%d D=%d bags (%.1f s), torch-CPU oracle, best of thread counts %s "
                                            "on a %d-thread host" % (n_cpu, N_PATCH, D_FEAT, el, sorted(probe), all_cores),
                                  "host_physical_cores": host_cores()[0], "host_logical_cpus": host_cores()[1],
                                  "probe_slides_per_s": {str(k): round(v, 2) for k, v in probe.items()},
                                  "ms_per_slide": round(1e3 / cpu_sps, 2)}
        result["speedup_vs_cpu"] = round(slides_per_s / cpu_sps, 1)
        result["max_abs_err_vs_oracle"] = err
    return result


if __name__ == "__main__":
    main()
