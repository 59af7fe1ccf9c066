"""Step3-style trainer for the MI355X aggregation path (`--arch ga | abmil | transmil`).

Own-code counterpart of the reference's `Step3_WSI_classification_ACMIL.py` (main :59-173,
train_one_epoch :175-235, evaluate :242-286) and of the helpers it takes from `utils/utils.py`
(Struct :246-248, adjust_learning_rate :250-262, save_model :415-422).  Same hyper-parameters, loss
assembly (sub-branch CE + bag CE + pairwise-cosine diversity loss), AdamW, per-iteration cosine schedule,
model selection on val_f1 + val_auc and checkpoint dictionary {'model','optimizer','epoch','config'}.

What is new (not in the reference, which is single-GPU): slide-level data parallelism -- one process per
GPU (`torchrun`), rank r takes slides r, r+G, ... of the shuffled epoch, and every step all-reduces ONE flat
fp32 gradient bucket (0.8 MB) with RCCL (`torch.distributed`, backend nccl; gloo on CPU for tests).

Data: bags are dicts {'input': [N,D] fp16/fp32, 'label': int} as produced by the reference's
`datasets/datasets.py:138-155`; this file reads `.npz` bag directories or generates synthetic bags (the
reference's HDF5 reader needs h5py, which is outside the hot path).
"""
from __future__ import annotations

import argparse
import math
import os
import random
import time
from typing import Dict, List, Optional, Sequence

import numpy as np
import torch
import torch.nn.functional as F

from . import ops
from .staging import staged, staged_groups, staged_train_groups

EVAL_BATCH = 64     # slides per fused eval launch (acmil_ga_forward_batch takes up to 64): the ragged last round of tiles and the
                    # merge / heads launches weigh ~2 % at 64 bags of 50 000 patches, ~9 % at 16

PRETRAIN_DIMS = {  # Step3_WSI_classification_ACMIL.py:69-87
    "medical_ssl": (384, 128), "natural_supervised": (512, 256), "path-clip-B": (512, 256), "openai-clip-B": (512, 256),
    "plip": (512, 256), "quilt-net": (512, 256), "path-clip-B-AAAI": (512, 256), "biomedclip": (512, 256),
    "path-clip-L-336": (768, 384), "openai-clip-L-336": (768, 384), "UNI": (1024, 512), "GigaPath": (1536, 768),
}


class Struct:  # utils/utils.py:246-248
    def __init__(self, **entries):
        self.__dict__.update(entries)


def set_seed(seed: int) -> None:  # utils/utils.py:226-244
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if torch.cuda.is_available():
        torch.cuda.manual_seed_all(seed)


def adjust_learning_rate(optimizer, epoch: float, cfg) -> float:
    """Linear warm-up then half-cycle cosine, evaluated per iteration (utils/utils.py:250-262)."""
    if epoch < cfg.warmup_epoch:
        lr = cfg.lr * epoch / cfg.warmup_epoch
    else:
        lr = cfg.min_lr + (cfg.lr - cfg.min_lr) * 0.5 * (
            1.0 + math.cos(math.pi * (epoch - cfg.warmup_epoch) / (cfg.train_epoch - cfg.warmup_epoch)))
    for group in optimizer.param_groups:
        group["lr"] = lr * group["lr_scale"] if "lr_scale" in group else lr
    return lr


def save_model(conf, epoch, model, optimizer, save_path) -> None:  # utils/utils.py:415-422
    torch.save({"model": model.state_dict(), "optimizer": optimizer.state_dict(), "epoch": epoch, "config": conf}, save_path)


# ----------------------------------------------------------------------------------------------- losses
def acmil_losses(sub_preds, slide_preds, attn, labels, n_token: int):
    """(loss0, loss1, diff_loss) of Step3_WSI_classification_ACMIL.py:201-214 (trainer-side, tiny tensors)."""
    if n_token > 1:
        loss0 = F.cross_entropy(sub_preds, labels.repeat_interleave(n_token))
    else:
        loss0 = torch.zeros((), device=slide_preds.device)
    loss1 = F.cross_entropy(slide_preds, labels)
    diff_loss = torch.zeros((), device=slide_preds.device)
    p = torch.softmax(attn, dim=-1)
    for i in range(n_token):
        for j in range(i + 1, n_token):
            diff_loss = diff_loss + torch.cosine_similarity(p[:, i], p[:, j], dim=-1).mean() / (n_token * (n_token - 1) / 2)
    return loss0, loss1, diff_loss


# ----------------------------------------------------------------------------------------------- metrics
def multiclass_auroc(prob: torch.Tensor, target: torch.Tensor, n_class: int) -> float:
    """Macro one-vs-rest AUROC (what torchmetrics.AUROC(task='multiclass') computes by default); classes absent
    from `target` are skipped."""
    prob, target = prob.detach().cpu().double(), target.detach().cpu()
    aucs = []
    for c in range(n_class):
        pos = target == c
        n_pos, n_neg = int(pos.sum()), int((~pos).sum())
        if n_pos == 0 or n_neg == 0:
            continue
        s = prob[:, c]
        order = torch.argsort(s)
        ranks = torch.empty_like(s)
        ranks[order] = torch.arange(1, len(s) + 1, dtype=torch.double)
        # average ranks over ties
        uniq, inv = torch.unique(s, return_inverse=True)
        if len(uniq) != len(s):
            sums = torch.zeros(len(uniq), dtype=torch.double).scatter_add_(0, inv, ranks)
            cnts = torch.zeros(len(uniq), dtype=torch.double).scatter_add_(0, inv, torch.ones_like(ranks))
            ranks = (sums / cnts)[inv]
        aucs.append(float((ranks[pos].sum() - n_pos * (n_pos + 1) / 2) / (n_pos * n_neg)))
    return float(np.mean(aucs)) if aucs else float("nan")


def micro_f1(prob: torch.Tensor, target: torch.Tensor) -> float:
    """torchmetrics.F1Score(task='multiclass') default (micro average) == accuracy of argmax."""
    return float((prob.argmax(dim=1).cpu() == target.cpu()).double().mean())


# ----------------------------------------------------------------------------------------------- data
class SyntheticBags:
    """Deterministic stand-in for HDF5_feat_dataset2: fp16 bags with a weak class signal so that training moves."""

    def __init__(self, n_slides: int, n_patches, d_feat: int, n_class: int, seed: int = 0):
        self.items = []
        g = torch.Generator().manual_seed(seed)
        for i in range(n_slides):
            n = n_patches if isinstance(n_patches, int) else int(torch.randint(n_patches[0], n_patches[1] + 1, (1,), generator=g))
            label = i % n_class
            x = torch.randn(n, d_feat, generator=g)
            k = max(1, n // 50)
            x[:k, label::n_class] += 1.5   # a few "tumour" patches carry the class signal
            self.items.append({"input": x.half(), "label": label})

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


class NpzBags:
    """Directory of `<slide>.npz` files with arrays `feat` [N,D] and `label` (same content as the reference's HDF5 groups)."""

    def __init__(self, root: str):
        self.files = sorted(os.path.join(root, f) for f in os.listdir(root) if f.endswith(".npz"))

    def __len__(self):
        return len(self.files)

    def __getitem__(self, i):
        z = np.load(self.files[i])
        return {"input": torch.from_numpy(z["feat"]), "label": int(z["label"])}


def epoch_order(n: int, epoch: int, seed: int, shuffle: bool, rank: int, world: int, drop_last: bool = True) -> List[int]:
    """Slide order of one epoch for this rank: the same permutation on every rank, strided by rank."""
    idx = list(range(n))
    if shuffle:
        rng = random.Random(seed * 100003 + epoch)
        rng.shuffle(idx)
    if world > 1:
        usable = (n // world) * world if drop_last else n
        idx = idx[:usable][rank::world]
    return idx


# ----------------------------------------------------------------------------------------------- data parallel
class GradBucket:
    """All parameter gradients as ONE flat fp32 buffer: a single all-reduce per step (latency-bound at 0.8 MB,
    so one collective beats per-tensor calls), averaged over ranks, scattered back as views."""

    def __init__(self, params: Sequence[torch.nn.Parameter]):
        self.params = [p for p in params if p.requires_grad]
        self.numel = sum(p.numel() for p in self.params)
        dev = self.params[0].device
        # + 1: the range flag of the step rides behind the gradients (optim.FlatAdamW, csrc/ga_step.hip): one all-reduce tells
        # every rank whether ANY rank's bag left the split-f16 range, and every rank's optimizer launch skips that step
        self.flat = torch.zeros(self.numel + 1, dtype=torch.float32, device=dev)
        self.flag = self.flat[self.numel:]
        self.peer = None        # peer.PeerReducer once enable_direct() succeeded: the reduction then happens inside the optimizer launch
        off = 0
        for p in self.params:   # gradients become views into the flat buffer: no copy in / out
            p.grad = self.flat[off:off + p.numel()].view_as(p)
            off += p.numel()

    def zero(self):
        self.flat.zero_()

    def sync_from_grads(self):
        """Re-point after an autograd pass that replaced .grad tensors.  (The fused step writes into the bucket views: then this is
        one pointer comparison per parameter -- building a view per parameter and step cost ~90 us of host time.)"""
        base = self.flat.data_ptr()
        off = 0
        for p in self.params:
            n = p.numel()
            g = p.grad
            if g is None or g.data_ptr() != base + 4 * off:
                view = self.flat[off:off + n].view_as(p)
                if g is None:
                    view.zero_()
                else:
                    view.copy_(g)
                p.grad = view
            off += n

    def enable_direct(self, rank: int, world: int, group=None) -> bool:
        """Switch to the one-shot direct reduction (acmil_amd/peer.py: the optimizer launch reads the peers' buckets through IPC-mapped
        pointers).  Collective; False (on every rank) when the mapping cannot be set up -- torch.distributed stays in charge."""
        if world > 1 and self.flat.is_cuda:
            from .peer import PeerReducer
            self.peer = PeerReducer.try_create(self.flat.numel(), self.flat.device, rank, world, group)
        return self.peer is not None

    def allreduce_mean(self, world: int):
        # the direct path replaces the collective ONLY once an optimizer has taken the reducer over (FlatAdamW sets `owner`): with any
        # other optimizer (conf.torch_optimizer) nobody would reduce at all and the ranks would silently diverge
        if self.peer is not None and getattr(self.peer, "owner", None) is not None:
            return                      # reduced by FlatAdamW.step itself (acmil_adamw_step_peer)
        if world > 1:
            import torch.distributed as dist
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            self.flat.div_(world)


def broadcast_parameters(model, world: int):
    if world > 1:
        import torch.distributed as dist
        for t in list(model.parameters()) + list(model.buffers()):
            dist.broadcast(t.data, src=0)


# ----------------------------------------------------------------------------------------------- loops
def train_one_epoch(model, data, optimizer, device, epoch: int, conf, bucket: Optional[GradBucket] = None,
                    rank: int = 0, world: int = 1, log_every: int = 100, fused: bool = True,
                    order: Optional[Sequence[int]] = None, uniforms_fn=None) -> Dict[str, float]:
    """One epoch, one slide per iteration per rank (Step3_WSI_classification_ACMIL.py:175-227).
    fused=True uses the module's autograd-free `train_step` (fused HIP loss + backward); fused=False runs the
    reference's op sequence through torch autograd on top of the HIP forward/backward node (same gradients).
    order / uniforms_fn (parity tests against a trajectory the reference's own loop produced): the slide order of this epoch instead of
    the shuffle, and uniforms_fn(epoch, it) -> the [K, k] STKIM draw of iteration `it` instead of the device draw."""
    if int(getattr(conf, "bags_per_step", 1) or 1) > 1:
        return train_one_epoch_groups(model, data, optimizer, device, epoch, conf, bucket, rank, world, log_every, fused)
    model.train()
    order = list(order) if order is not None else epoch_order(len(data), epoch, conf.seed, True, rank, world)
    sums = {"sub_loss": 0.0, "diff_loss": 0.0, "slide_loss": 0.0}
    acc = torch.zeros(4, device=device)          # loss sums stay on the device: no .item() sync per step
    use_fused = fused and hasattr(model, "train_step")
    t0 = time.time()
    label_dev = torch.arange(conf.n_class, device=device)      # labels are picked on the device: no H2D per step
    # Range guard of the split-f16 step without a host read-back per step: the step leaves a flag on the device, the optimizer
    # launch skips a flagged step, and the host looks at flags two steps late; a skipped bag is then repeated in fp32 arithmetic
    # (what the reference computes), two positions later in the epoch's order.
    lagged = (use_fused and getattr(optimizer, "guard_flag", None) is not None and getattr(model, "range_guard", False)
              and getattr(model, "precision", "") == "f16x3")
    recent: Dict[int, tuple] = {}
    opt_aware = use_fused and hasattr(optimizer, "in_step_args") and getattr(model, "supports_in_step_optimizer", False)
    in_step = opt_aware and lagged and world == 1

    def reduce_and_step(track):
        if bucket is not None:
            bucket.sync_from_grads()
            bucket.allreduce_mean(world)
        return optimizer.step(track_flag=True) if track else optimizer.step()

    def settle(lag):
        nonlocal acc
        limit = optimizer._step_id - lag
        # tracked steps <= limit whose flag the device saw set (a timed-out peer wait of the direct reduction raises here, two steps late)
        skipped = optimizer.poll_skipped(lag)
        for sid in [k for k in sorted(recent) if k <= limit]:
            idx, ls = recent.pop(sid)
            if sid not in skipped:
                acc += ls
                continue
            item = data[idx]                                     # the staging ring has long recycled the bag: read it again
            xs = torch.as_tensor(item["input"]).to(device)
            ys = label_dev[item["label"]:item["label"] + 1]
            redo, _ = model.train_step(xs.unsqueeze(0), ys, precision="fp32", guard_flag=optimizer.guard_flag)
            reduce_and_step(False)
            acc += redo

    for it, item in enumerate(staged(data, order, device)):     # pinned double-buffered H2D on a copy stream (staging.py)
        x = item["input"]                                      # fp16 stays fp16: converted inside the kernel
        labels = label_dev[item["label"]:item["label"] + 1]
        adjust_learning_rate(optimizer, epoch + it / len(order), conf)
        sid = None
        if use_fused:
            # single GPU: the step applies the optimizer itself (its closing launch = gradient finish + AdamW + weight re-pack) where it can
            losses, out = model.train_step(x.unsqueeze(0), labels, guard_flag=optimizer.guard_flag if lagged else None,
                                           **({"uniforms": uniforms_fn(epoch, it)} if uniforms_fn is not None else {}),
                                           **({"optimizer": optimizer, "track_flag": True, "in_step": in_step} if opt_aware else {}))
            sid = out.get("opt_step_id") if in_step else None
            if not lagged:
                acc += losses
        else:
            out = model(x.unsqueeze(0) if x.dtype == torch.float32 or conf.arch == "ga" else x.float().unsqueeze(0))
            if isinstance(out, tuple):                       # ACMIL: (sub_preds, slide_preds, attn)
                sub_preds, slide_preds, attn = out
                loss0, loss1, diff_loss = acmil_losses(sub_preds, slide_preds, attn, labels, conf.n_token)
            else:                                            # single-head models (abmil, transmil): criterion(output, label),
                zero = torch.zeros((), device=device)        # Step3_WSI_classification.py / engine.py:19-21
                loss0, loss1, diff_loss = zero, F.cross_entropy(out, labels), zero
            loss = diff_loss + loss0 + loss1
            optimizer.zero_grad(set_to_none=False)
            loss.backward()
            acc += torch.stack([loss0.detach(), loss1.detach(), diff_loss.detach(), loss.detach()])
        if sid is None:
            sid = reduce_and_step(lagged)
        if lagged:
            recent[sid] = (item["index"], losses)
            settle(2)
        if log_every and (it + 1) % log_every == 0 and bucket is not None and bucket.peer is not None:
            bucket.peer.check()          # untracked steps (no lagged flag): look at the reducer's timeout word where the loop syncs anyway
        if rank == 0 and log_every and (it + 1) % log_every == 0:
            a = acc.tolist()
            sums = {"sub_loss": a[0], "slide_loss": a[1], "diff_loss": a[2]}
        if rank == 0 and log_every and (it + 1) % log_every == 0:
            print("Epoch: [%d] [%d/%d] lr: %.6f sub_loss: %.4f diff_loss: %.4f slide_loss: %.4f (%.1f slides/s/rank)" % (
                epoch, it + 1, len(order), optimizer.param_groups[0]["lr"], sums["sub_loss"] / (it + 1),
                sums["diff_loss"] / (it + 1), sums["slide_loss"] / (it + 1), (it + 1) / (time.time() - t0)))
    if lagged:
        settle(0)
    if bucket is not None and bucket.peer is not None:
        bucket.peer.check()              # a peer that never published leaves this rank's updates unapplied: raise, never train on silently
    n = max(1, len(order))
    a = acc.tolist()
    return {"sub_loss": a[0] / n, "slide_loss": a[1] / n, "diff_loss": a[2] / n}


def train_one_epoch_groups(model, data, optimizer, device, epoch: int, conf, bucket: Optional[GradBucket] = None,
                           rank: int = 0, world: int = 1, log_every: int = 100, fused: bool = True) -> Dict[str, float]:
    """One epoch with conf.bags_per_step = G > 1 slides per step and rank: every step runs on the MEAN gradient of its G slides
    (ACMIL_GA.train_step_batch -> acmil_ga_train_step_group: the per-slide forward / losses / backward of
    Step3_WSI_classification_ACMIL.py:189-221, batched), i.e. what G data-parallel ranks compute -- with `world` ranks the step
    averages G x world slides and all-reduces ONCE per G slides.  Not the reference's B = 1 SGD trajectory (neither is its DP
    twin): opt-in.  The per-iteration cosine schedule advances per step; the epoch has ceil(len / G) steps per rank.
    Models without a group step (autograd fallback, CPU tests) accumulate the G slides' gradients one by one."""
    model.train()
    G = int(conf.bags_per_step)
    order = epoch_order(len(data), epoch, conf.seed, True, rank, world)
    acc = torch.zeros(4, device=device)
    use_fused = fused and hasattr(model, "train_step_batch") and device.type == "cuda"
    lagged = (use_fused and getattr(optimizer, "guard_flag", None) is not None and getattr(model, "range_guard", False)
              and getattr(model, "precision", "") == "f16x3")
    opt_aware = use_fused and hasattr(optimizer, "in_step_args") and getattr(model, "supports_in_step_optimizer", False)
    in_step = opt_aware and lagged and world == 1
    recent: Dict[int, tuple] = {}
    label_dev = torch.arange(conf.n_class, device=device)
    t0 = time.time()
    n_steps = (len(order) + G - 1) // G

    def reduce_and_step(track):
        if bucket is not None:
            bucket.sync_from_grads()
            bucket.allreduce_mean(world)
        return optimizer.step(track_flag=True) if track else optimizer.step()

    def settle(lag):
        nonlocal acc
        limit = optimizer._step_id - lag
        skipped = optimizer.poll_skipped(lag)
        for sid in [k for k in sorted(recent) if k <= limit]:
            idxs, labs, ls = recent.pop(sid)
            if sid not in skipped:
                acc += ls
                continue
            # a bag of the group left the split-f16 range: the whole group again in exact fp32 arithmetic (bag by bag, mean gradient)
            xs = [torch.as_tensor(data[i]["input"]).to(device) for i in idxs]
            redo, _ = model.train_step_batch(xs, label_dev[torch.tensor(labs, device=device)], precision="fp32", guard_flag=optimizer.guard_flag)
            reduce_and_step(False)
            acc += redo.sum(0)

    for it, grp in enumerate(staged_train_groups(data, order, device, G)):
        labels = label_dev[torch.tensor(grp["labels"], device=device)] if device.type == "cuda" else torch.tensor(grp["labels"])
        adjust_learning_rate(optimizer, epoch + it / max(1, n_steps), conf)
        sid = None
        if use_fused:
            losses, out = model.train_step_batch((grp["input"], grp["rows"]), labels, guard_flag=optimizer.guard_flag if lagged else None,
                                                 **({"optimizer": optimizer, "track_flag": True, "in_step": in_step} if opt_aware else {}))
            sid = out.get("opt_step_id") if in_step else None
            ls = losses.sum(0)
            if not lagged:
                acc += ls
        else:
            optimizer.zero_grad(set_to_none=False)
            ls = torch.zeros(4, device=device)
            off = 0
            for b, n in enumerate(grp["rows"]):
                x = grp["input"][off:off + n]
                off += n
                y = labels[b:b + 1]
                out = model(x.unsqueeze(0) if x.dtype == torch.float32 or conf.arch == "ga" else x.float().unsqueeze(0))
                if isinstance(out, tuple):
                    loss0, loss1, diff_loss = acmil_losses(out[0], out[1], out[2], y, conf.n_token)
                else:
                    zero = torch.zeros((), device=device)
                    loss0, loss1, diff_loss = zero, F.cross_entropy(out, y), zero
                loss = diff_loss + loss0 + loss1
                (loss / len(grp["rows"])).backward()          # .grad accumulates: the group's mean gradient
                ls = ls + torch.stack([loss0.detach(), loss1.detach(), diff_loss.detach(), loss.detach()])
            acc += ls
        if sid is None:
            sid = reduce_and_step(lagged)
        if lagged:
            recent[sid] = (grp["indices"], grp["labels"], ls)
            settle(2)
        if rank == 0 and log_every and (it + 1) % log_every == 0:
            a = acc.tolist()
            seen = min(len(order), (it + 1) * G)
            print("Epoch: [%d] [%d/%d] lr: %.6f sub_loss: %.4f diff_loss: %.4f slide_loss: %.4f (%.1f slides/s/rank, %d bags per step)" % (
                epoch, it + 1, n_steps, optimizer.param_groups[0]["lr"], a[0] / seen, a[2] / seen, a[1] / seen, seen / (time.time() - t0), G))
    if lagged:
        settle(0)
    if bucket is not None and bucket.peer is not None:
        bucket.peer.check()
    n = max(1, len(order))
    a = acc.tolist()
    return {"sub_loss": a[0] / n, "slide_loss": a[1] / n, "diff_loss": a[2] / n}


@torch.no_grad()
def evaluate(model, data, device, conf, header: str = "Val", rank: int = 0, world: int = 1, batched: bool = True, detail: Optional[dict] = None):
    """(auroc, acc, f1, loss) as Step3_WSI_classification_ACMIL.py:242-286; slides sharded over ranks, gathered on all.
    batched=False keeps the reference's one-slide-per-call pattern (`model(x)` per slide); detail: a dict that receives this
    rank's per-slide 'prob' [n,C], 'loss' [n] and 'div' [n] (tests)."""
    model.eval()
    order = epoch_order(len(data), 0, 0, False, rank, world, drop_last=False)
    probs, labels, losses, divs = [], [], [], []
    label_dev = torch.arange(conf.n_class, device=device)
    on_gpu = batched and device.type == "cuda" and hasattr(model, "forward_batch")
    fused_batch = on_gpu and (getattr(model, "_is_fused", lambda: False)() or getattr(model, "_is_wide_fused", lambda: False)())
    composed_group = on_gpu and not fused_batch and getattr(model, "_is_composed_groupable", lambda p: False)(getattr(model, "precision", ""))
    if fused_batch or composed_group:
        # GA at the fused widths: up to EVAL_BATCH = 64 staged bags share ONE fused launch (acmil_ga_forward_batch -- the launch bench.py times).
        # GA at the composed widths (GigaPath 1536 -> 768, n_token > 5): groups of <= 16 slides are staged with their rows back to back
        # (staged_train_groups: H2D straight into one ring slot) and go through ONE projection launch, ONE gated-score launch and a
        # per-bag pooling pass (ACMIL_GA.forward_group).  Either way the split-f16 range word of a batch is looked at only after the
        # NEXT batch has been enqueued, so the GPU never idles on the check; a flagged batch (never seen on real features) is re-read
        # and repeated in fp32 arithmetic.
        n_total = len(order)
        probs, labels, losses, divs = [None] * n_total, [None] * n_total, [None] * n_total, [None] * n_total
        pos = {idx: j for j, idx in enumerate(order)}

        def record(item_idx, label, triple):
            _, slide_preds, attn = triple
            j = pos[item_idx]
            y = label_dev[label:label + 1]
            divs[j] = ops.attn_entropy_loss(attn)
            losses[j] = F.cross_entropy(slide_preds, y)
            probs[j] = torch.softmax(slide_preds, dim=-1)
            labels[j] = y

        def settle(pending):
            status, members = pending
            if status is None or int(status) == 0:
                return
            model._fb_host = getattr(model, "_fb_host", 0) + 1          # host-decided repeat (see _GatedBase.range_fallbacks)
            xs = [torch.as_tensor(data[i]["input"]).to(device) for i, _ in members]     # the ring has recycled them: read again
            for (i, lab), triple in zip(members, model.forward_batch(xs, precision="fp32")):
                record(i, lab, triple)

        def batches():      # (members [(index, label)], triples, status) per launch group
            if fused_batch:
                for group in staged_groups(data, order, device, group=EVAL_BATCH):
                    triples, status = model.forward_batch([it["input"] for it in group], defer_guard=True)
                    yield [(it["index"], it["label"]) for it in group], triples, status
            else:
                for g in staged_train_groups(data, order, device, group=ops.MAX_GROUP):
                    triples, status = model.forward_group(g["input"], g["rows"], defer_guard=True)
                    yield list(zip(g["indices"], g["labels"])), triples, status

        pending = None
        for members, triples, status in batches():
            for (i, lab), triple in zip(members, triples):
                record(i, lab, triple)
            if pending is not None:
                settle(pending)
            pending = (status, members)
        if pending is not None:
            settle(pending)
    else:
        for item in staged(data, order, device):
            x = item["input"]
            y = label_dev[item["label"]:item["label"] + 1]
            out = model(x.unsqueeze(0) if x.dtype == torch.float32 or conf.arch == "ga" else x.float().unsqueeze(0))
            if isinstance(out, tuple):
                sub_preds, slide_preds, attn = out
                # per-slide scalars stay on the device (a float() here would serialise the H2D of the next bag with this forward)
                divs.append(ops.attn_entropy_loss(attn))          # div_loss (:259) from one HIP pass over the raw scores
            else:
                slide_preds = out
            losses.append(F.cross_entropy(slide_preds, y))
            probs.append(torch.softmax(slide_preds, dim=-1))
            labels.append(y)
    prob = torch.cat(probs) if probs else torch.zeros(0, conf.n_class, device=device)
    lab = torch.cat(labels) if labels else torch.zeros(0, dtype=torch.long, device=device)
    if detail is not None:
        detail.update(prob=prob.cpu(), loss=torch.stack(losses).cpu() if losses else torch.zeros(0),
                      div=torch.stack([d.reshape(()) for d in divs]).cpu() if divs else torch.zeros(0))
    losses = torch.stack(losses).tolist() if losses else []
    if world > 1:
        import torch.distributed as dist
        gathered = [None] * world
        dist.all_gather_object(gathered, (prob.cpu(), lab.cpu(), losses))
        prob = torch.cat([g[0] for g in gathered]); lab = torch.cat([g[1] for g in gathered])
        losses = [l for g in gathered for l in g[2]]
    auroc = multiclass_auroc(prob, lab, conf.n_class)
    acc = float((prob.argmax(1).cpu() == lab.cpu()).double().mean()) * 100.0
    f1 = micro_f1(prob, lab)
    loss = float(np.mean(losses)) if losses else float("nan")
    if rank == 0:
        print("* %s Acc@1 %.3f loss %.3f auroc %.3f f1_score %.3f" % (header, acc, loss, auroc, f1))
    return auroc, acc, f1, loss


def make_optimizer(model, conf, device, bucket: Optional["GradBucket"] = None, lr: float = 0.001):
    """AdamW as in Step3_WSI_classification_ACMIL.py:139.  On the GPU: acmil_amd.optim.FlatAdamW (same update rule, parameters /
    gradients / moments in flat buffers, ONE launch per step; shares the DP gradient bucket); on CPU (gloo tests): torch's."""
    params = [p for p in model.parameters() if p.requires_grad]
    if device.type == "cuda" and not getattr(conf, "torch_optimizer", False):
        from .optim import FlatAdamW
        opt = FlatAdamW(params, lr=lr, weight_decay=conf.wd, grad_buffer=None if bucket is None else bucket.flat,
                        on_step=getattr(model, "invalidate_packed", None), peer=None if bucket is None else bucket.peer)
        if getattr(conf, "arch", "") == "ga" and getattr(conf, "n_token", 2) == 1:
            # n_token == 1: the branch-head loss is not built (Step3_WSI_classification_ACMIL.py:201-204), its parameters keep
            # grad None in the reference and torch's AdamW leaves them untouched (no weight decay)
            opt.set_frozen(list(model.classifier[0].parameters()))
        if hasattr(model, "adamw_pack_hook"):
            opt.pack_hook = model.adamw_pack_hook(opt)      # the optimizer launch also re-packs the module's weights (acmil_ga_adamw_pack)
        return opt
    if bucket is not None and bucket.peer is not None:
        # the direct reduction lives inside FlatAdamW's launch: with torch's optimizer the bucket goes back to the collective
        print("acmil_amd.train: --dp-reduce direct needs FlatAdamW; using torch.distributed all_reduce with torch.optim.AdamW")
        bucket.peer.close()         # collective (every rank takes this path): drain, barrier, unmap -- peers still hold the slots mapped
        bucket.peer = None
    return torch.optim.AdamW(params, lr=lr, weight_decay=conf.wd)


def build_model(conf):
    from .architecture.transformer import ABMIL, ACMIL_GA
    if conf.arch == "ga":
        return ACMIL_GA(conf, n_token=conf.n_token, n_masked_patch=conf.n_masked_patch, mask_drop=conf.mask_drop,
                        precision=conf.precision)
    if conf.arch == "abmil":
        return ABMIL(conf, precision=conf.precision)
    if conf.arch == "transmil":
        from .architecture.transMIL import TransMIL
        return TransMIL(conf)
    if conf.arch == "mha":      # Step3_WSI_classification_ACMIL.py:127-128
        from .architecture.transformer import ACMIL_MHA
        return ACMIL_MHA(conf, n_token=conf.n_token, n_masked_patch=conf.n_masked_patch, mask_drop=conf.mask_drop,
                         precision="fp32" if conf.precision == "fp32" else "f16x3")
    raise SystemExit("--arch %s is not on the MI355X path yet (ga, abmil, mha, transmil)" % conf.arch)


def get_arguments(argv=None):
    p = argparse.ArgumentParser("WSI classification training (MI355X aggregation path)")
    p.add_argument("--config", default=None, help="yaml with the reference's keys (train_epoch, warmup_epoch, wd, lr, min_lr, n_class, ...)")
    p.add_argument("--seed", type=int, default=1)
    p.add_argument("--n_token", type=int, default=1)
    p.add_argument("--n_masked_patch", type=int, default=0)
    p.add_argument("--mask_drop", type=float, default=0.6)
    p.add_argument("--arch", default="ga", choices=["ga", "abmil", "mha", "transmil"])
    p.add_argument("--pretrain", default="medical_ssl", choices=sorted(PRETRAIN_DIMS))
    p.add_argument("--lr", type=float, default=1e-4)
    p.add_argument("--precision", default="f16x3", choices=["f16x3", "fp32", "f16"])
    p.add_argument("--data_dir", default=None, help="directory with train/ val/ test/ sub-directories of .npz bags; synthetic if omitted")
    p.add_argument("--synthetic_slides", type=int, default=64)
    p.add_argument("--synthetic_patches", type=int, default=10000)
    p.add_argument("--train_epoch", type=int, default=None)
    p.add_argument("--n_class", type=int, default=None)
    p.add_argument("--out_dir", default="runs/acmil")
    p.add_argument("--bags-per-step", dest="bags_per_step", type=int, default=1,
                   help="slides per optimizer step and rank: > 1 trains on the MEAN gradient of that many slides per step (one batched "
                        "group step on the GPU; under data parallelism one all-reduce per group) -- the throughput mode; 1 = the "
                        "reference's B = 1 SGD (Step3_WSI_classification_ACMIL.py:189-221)")
    p.add_argument("--dp-reduce", dest="dp_reduce", default=os.environ.get("ACMIL_DP_REDUCE", "rccl"), choices=["rccl", "direct"],
                   help="data-parallel gradient reduction: torch.distributed all_reduce (RCCL), or the one-shot direct reduction fused "
                        "into the optimizer launch (peers' buckets read through IPC-mapped pointers; one node; falls back to rccl)")
    return p.parse_args(argv)


def main(argv=None):
    args = get_arguments(argv)
    c = dict(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, dataset="synthetic", B=1, n_class=2, data_dir=None, config=None)
    if args.config:
        import yaml
        with open(args.config) as f:
            c.update(yaml.safe_load(f))
    c.update({k: v for k, v in vars(args).items() if v is not None})
    conf = Struct(**c)
    conf.D_feat, conf.D_inner = PRETRAIN_DIMS[conf.pretrain]

    world, rank = int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    device = torch.device("cuda", local_rank)
    torch.cuda.set_device(device)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=device)
    set_seed(conf.seed)
    if conf.data_dir:
        train, val, test = (NpzBags(os.path.join(conf.data_dir, s)) for s in ("train", "val", "test"))
    else:
        mk = lambda n, s: SyntheticBags(n, conf.synthetic_patches, conf.D_feat, conf.n_class, seed=s)
        train, val, test = mk(conf.synthetic_slides, 1), mk(max(8, conf.synthetic_slides // 4), 2), mk(max(8, conf.synthetic_slides // 4), 3)
    model = build_model(conf).to(device)
    broadcast_parameters(model, world)
    bucket = GradBucket(list(model.parameters())) if world > 1 else None
    if bucket is not None and getattr(conf, "dp_reduce", "rccl") == "direct" and device.type == "cuda":
        ok = bucket.enable_direct(rank, world)
        if rank == 0:
            print("data-parallel gradient reduction: %s" % ("direct (fused into the optimizer launch)" if ok else "torch.distributed (direct unavailable)"))
    optimizer = make_optimizer(model, conf, device, bucket)
    os.makedirs(conf.out_dir, exist_ok=True)
    best = {"epoch": -1, "val_acc": 0, "val_auc": 0, "val_f1": 0, "test_acc": 0, "test_auc": 0, "test_f1": 0}
    for epoch in range(conf.train_epoch):
        train_one_epoch(model, train, optimizer, device, epoch, conf, bucket, rank, world, fused=hasattr(model, "train_step"))
        val_auc, val_acc, val_f1, _ = evaluate(model, val, device, conf, "Val", rank, world)
        test_auc, test_acc, test_f1, _ = evaluate(model, test, device, conf, "Test", rank, world)
        if val_f1 + val_auc > best["val_f1"] + best["val_auc"]:
            best.update(epoch=epoch, val_auc=val_auc, val_acc=val_acc, val_f1=val_f1, test_auc=test_auc, test_acc=test_acc, test_f1=test_f1)
            if rank == 0:
                save_model(conf, epoch, model, optimizer, os.path.join(conf.out_dir, "checkpoint-best.pth"))
    if rank == 0:
        save_model(conf, epoch, model, optimizer, os.path.join(conf.out_dir, "checkpoint-last.pth"))
        print("Results on best epoch:"); print(best)
    if bucket is not None and bucket.peer is not None:
        bucket.peer.close()          # collective: a peer's last optimizer launch may still be reading this rank's slots
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
