"""CPU tests of the host-side trainer logic (no GPU): schedule, metrics, sharding, checkpoint format, and the
N>1 data-parallel gradient path on 2 gloo processes (all-reduced bucket == mean of the per-rank gradients)."""
import os
import socket
import tempfile

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from acmil_amd import train as T
from oracle import ga_oracle as O


def test_lr_schedule_matches_oracle_restatement():
    cfg = T.Struct(lr=1e-3, min_lr=1e-5, warmup_epoch=2, train_epoch=20)
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=0.1)
    for e in [0.0, 0.5, 1.99, 2.0, 7.3, 19.99]:
        lr = T.adjust_learning_rate(opt, e, cfg)
        assert lr == pytest.approx(O.adjust_learning_rate(e, 1e-3, 1e-5, 2, 20))
        assert opt.param_groups[0]["lr"] == lr


def test_losses_match_oracle_restatement():
    g = torch.Generator().manual_seed(0)
    sub, slide, attn = torch.randn(5, 3, generator=g), torch.randn(1, 3, generator=g), torch.randn(1, 5, 400, generator=g)
    y = torch.tensor([2])
    a = T.acmil_losses(sub, slide, attn, y, 5)
    b = O.acmil_losses(sub, slide, attn, y, 5)
    for u, v in zip(a, b):
        assert float(u) == pytest.approx(float(v), abs=1e-7)
    assert float(T.acmil_losses(sub[:1], slide, attn[:, :1], y, 1)[0]) == 0.0


def test_metrics():
    prob = torch.tensor([[0.9, 0.1], [0.8, 0.2], [0.3, 0.7], [0.4, 0.6], [0.6, 0.4]])
    y = torch.tensor([0, 0, 1, 1, 1])
    assert T.multiclass_auroc(prob, y, 2) == pytest.approx(1.0)       # class-1 scores 0.7,0.6,0.4 vs 0.1,0.2 -> perfect
    assert T.micro_f1(prob, y) == pytest.approx(0.8)
    from sklearn.metrics import roc_auc_score
    g = torch.Generator().manual_seed(1)
    p = torch.softmax(torch.randn(200, 4, generator=g), dim=1)
    t = torch.randint(0, 4, (200,), generator=g)
    assert T.multiclass_auroc(p, t, 4) == pytest.approx(roc_auc_score(t.numpy(), p.numpy(), multi_class="ovr", average="macro"), abs=1e-9)


def test_epoch_order_shards_are_disjoint_and_cover():
    for world in (1, 2, 4, 8):
        parts = [T.epoch_order(37, 3, 5, True, r, world) for r in range(world)]
        flat = [i for p in parts for i in p]
        assert len(set(flat)) == len(flat) == (37 // world) * world
        assert len({len(p) for p in parts}) == 1
    assert T.epoch_order(10, 0, 1, True, 0, 1) != T.epoch_order(10, 1, 1, True, 0, 1)
    full = [T.epoch_order(10, 0, 0, False, r, 3, drop_last=False) for r in range(3)]
    assert sorted(i for p in full for i in p) == list(range(10))


def test_checkpoint_dictionary_format():
    m = torch.nn.Linear(4, 2)
    opt = torch.optim.AdamW(m.parameters(), lr=1e-3)
    conf = T.Struct(lr=1e-4, n_token=5)
    with tempfile.TemporaryDirectory() as d:
        path = os.path.join(d, "checkpoint-last.pth")
        T.save_model(conf, 7, m, opt, path)
        ck = torch.load(path, weights_only=False)
    assert set(ck) == {"model", "optimizer", "epoch", "config"} and ck["epoch"] == 7 and ck["config"].n_token == 5


def test_synthetic_bags_are_deterministic_fp16():
    a, b = T.SyntheticBags(4, 100, 384, 3, seed=2), T.SyntheticBags(4, 100, 384, 3, seed=2)
    assert a[1]["input"].dtype == torch.float16 and a[1]["input"].shape == (100, 384)
    assert torch.equal(a[3]["input"], b[3]["input"]) and [x["label"] for x in a.items] == [0, 1, 2, 0]


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _dp_worker(rank, world, port, out):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)                      # identical parameters on every rank
    model = torch.nn.Sequential(torch.nn.Linear(6, 5, bias=False), torch.nn.Linear(5, 3))
    T.broadcast_parameters(model, world)
    bucket = T.GradBucket(list(model.parameters()))
    opt = torch.optim.AdamW(model.parameters(), lr=1e-2)
    g = torch.Generator().manual_seed(100 + rank)      # a different "slide" per rank
    x, y = torch.randn(7, 6, generator=g), torch.randint(0, 3, (7,), generator=g)
    opt.zero_grad(set_to_none=False)
    torch.nn.functional.cross_entropy(model(x), y).backward()
    local = [p.grad.clone() for p in model.parameters()]
    bucket.sync_from_grads()
    bucket.allreduce_mean(world)
    reduced = [p.grad.clone() for p in model.parameters()]
    opt.step()
    gathered = [None] * world
    dist.all_gather_object(gathered, [t.numpy() for t in local])
    if rank == 0:
        torch.save({"reduced": reduced, "locals": gathered, "params": [p.detach().clone() for p in model.parameters()],
                    "numel": bucket.numel}, out)
    # every rank ends the step with identical parameters
    chk = [None] * world
    dist.all_gather_object(chk, [p.detach().numpy() for p in model.parameters()])
    assert all(np.array_equal(a, b) for a, b in zip(chk[0], chk[-1]))
    dist.destroy_process_group()


def test_data_parallel_gradient_bucket_gloo_world2():
    world, port = 2, _free_port()
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "r0.pt")
        mp.spawn(_dp_worker, args=(world, port, out), nprocs=world, join=True)
        r = torch.load(out, weights_only=False)
    assert r["numel"] == 6 * 5 + 5 * 3 + 3
    for i, red in enumerate(r["reduced"]):
        mean = sum(torch.from_numpy(loc[i]) for loc in r["locals"]) / world
        assert torch.allclose(red, mean, atol=1e-7)
        assert not torch.allclose(torch.from_numpy(r["locals"][0][i]), torch.from_numpy(r["locals"][1][i]))


# ---------------------------------------------------------------------------------- bench.py launcher (no GPU needed)
def test_bench_self_launch_command_and_dry_run_world2():
    """`python bench.py --gpus 2` without a launcher environment re-executes itself under torch.distributed.run (one rank per
    GPU, rendezvous on 127.0.0.1); --dry-run drives launcher, rendezvous, barrier and the max-over-ranks reduction on CPU/gloo."""
    import json, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import bench
    cmd = bench.self_launch_cmd(4, ["--gpus", "4", "--steps", "3"], port=29511)
    assert cmd[1:3] == ["-m", "torch.distributed.run"] and "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[-4:] == ["--gpus", "4", "--steps", "3"]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    for workload in ("ga_eval", "train"):
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--dry-run", "--steps", "3", "--warmup", "1",
                            "--workload", workload], env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
        js = json.loads(line)
        assert js["dry_run"] is True and js["n_gpus"] == 2 and js["steps"] == 3 and js["value"] is None
        if workload == "train":      # the data-parallel fields of the train line: the collective's time, and the direct reduction with its check
            assert js["allreduce_us"] > 0 and js["allreduce_bytes"] == 833216
            assert set(js["direct_reduce"]) >= {"ms_per_step", "first_step_check", "slot_memory"}


def _dp_step_worker(rank, world, port, out):
    """One data-parallel optimizer step wired exactly as bench.py --workload train / train.main: GradBucket first, the optimizer
    from make_optimizer(..., bucket), gradients written INTO the bucket views, sync_from_grads, allreduce_mean, step."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = torch.nn.Sequential(torch.nn.Linear(6, 5, bias=False), torch.nn.Tanh(), torch.nn.Linear(5, 3))
    T.broadcast_parameters(model, world)
    conf = T.Struct(wd=1e-2)
    bucket = T.GradBucket(list(model.parameters()))
    opt = T.make_optimizer(model, conf, torch.device("cpu"), bucket, lr=1e-2)
    views = [p.grad for p in model.parameters()]
    assert all(v.data_ptr() >= bucket.flat.data_ptr() for v in views)
    for step in range(3):
        grads = []
        for r in range(2):        # the gradients two ranks would produce (every process can compute both)
            g = torch.Generator().manual_seed(1000 * step + r)
            x, y = torch.randn(7, 6, generator=g), torch.randint(0, 3, (7,), generator=g)
            grads.append(torch.autograd.grad(torch.nn.functional.cross_entropy(model(x), y), list(model.parameters())))
        mine = grads[rank] if world > 1 else [(a + b) / 2 for a, b in zip(*grads)]
        for v, gnew in zip(views, mine):      # a fused train_step writes straight into the bucket views
            v.copy_(gnew)
        bucket.sync_from_grads()
        assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(model.parameters(), views))
        bucket.allreduce_mean(world)
        opt.step()
    if rank == 0:
        torch.save([p.detach().clone() for p in model.parameters()], out)
    if world > 1:
        chk = [None] * world
        dist.all_gather_object(chk, [p.detach().numpy() for p in model.parameters()])
        assert all(np.array_equal(a, b) for a, b in zip(chk[0], chk[-1]))
        dist.destroy_process_group()


def test_data_parallel_step_through_bucket_and_make_optimizer_matches_single_process():
    """2 gloo ranks, each with its own slide, end with identical parameters equal to ONE process stepping on the averaged gradients."""
    with tempfile.TemporaryDirectory() as d:
        o2, o1 = os.path.join(d, "w2.pt"), os.path.join(d, "w1.pt")
        mp.spawn(_dp_step_worker, args=(2, _free_port(), o2), nprocs=2, join=True)
        _dp_step_worker(0, 1, _free_port(), o1)
        a, b = torch.load(o2, weights_only=False), torch.load(o1, weights_only=False)
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, atol=1e-6, rtol=1e-6)


def _peer_fail_worker(rank, world, port, fail_rank):
    """PeerReducer.try_create where the set-up cannot succeed (no GPU here; `fail_rank` additionally fails EARLY, before its
    allocation): every rank must come back with None from the same number of collectives -- no hang, no mismatched collective."""
    import torch.distributed as dist
    from acmil_amd import peer as P
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    if rank == fail_rank:
        orig = P.PeerReducer._alloc
        P.PeerReducer._alloc = lambda self, memory: (_ for _ in ()).throw(RuntimeError("injected allocation failure"))
    red = P.PeerReducer.try_create(1000, torch.device("cpu"), rank, world)
    assert red is None
    t = torch.tensor([rank + 1.0])
    dist.all_reduce(t)                     # the process group is still in step: the next collective matches on every rank
    assert t.item() == 3.0
    # torch optimizer + a reducer nobody took over: the bucket keeps using the collective, make_optimizer drops the reducer
    from acmil_amd import train as T
    model = torch.nn.Linear(4, 3)
    bucket = T.GradBucket(list(model.parameters()))

    class _FakePeer:
        owner = None
        closed = 0

        def close(self):          # (collective in the real reducer: drain, barrier, unmap)
            _FakePeer.closed += 1
    bucket.peer = _FakePeer()
    bucket.flat.fill_(float(rank))
    bucket.allreduce_mean(world)
    assert torch.allclose(bucket.flat, torch.full_like(bucket.flat, 0.5))
    opt = T.make_optimizer(model, T.Struct(wd=0.0, torch_optimizer=True), torch.device("cpu"), bucket, lr=1e-3)
    assert isinstance(opt, torch.optim.AdamW) and bucket.peer is None
    assert _FakePeer.closed == 1           # dropped through its collective close(), not just forgotten
    dist.destroy_process_group()


@pytest.mark.parametrize("fail_rank", [-1, 1])
def test_peer_reducer_setup_failure_is_agreed_by_all_ranks(fail_rank):
    mp.spawn(_peer_fail_worker, args=(2, _free_port(), fail_rank), nprocs=2, join=True)


def test_grad_bucket_sync_repoints_only_what_autograd_replaced():
    """GradBucket.sync_from_grads: gradients that still are the bucket's views are left alone (one pointer comparison each -- the
    fused step writes into them); a .grad an autograd pass replaced is copied into the bucket and re-pointed; a missing one is zeroed."""
    from acmil_amd import train as T
    torch.manual_seed(0)
    lin1, lin2 = torch.nn.Linear(6, 4), torch.nn.Linear(4, 3)
    params = list(lin1.parameters()) + list(lin2.parameters())
    bucket = T.GradBucket(params)
    views = [p.grad for p in params]
    bucket.flat[:bucket.numel].copy_(torch.arange(bucket.numel, dtype=torch.float32))
    before = bucket.flat.clone()
    bucket.sync_from_grads()
    assert all(p.grad is v for p, v in zip(params, views)) and torch.equal(bucket.flat, before)       # nothing to do: nothing touched
    params[1].grad = torch.full_like(params[1], 7.0)          # what autograd does: a fresh tensor
    params[2].grad = None
    bucket.sync_from_grads()
    off = [0]
    for p in params:
        off.append(off[-1] + p.numel())
    assert all(p.grad.data_ptr() == bucket.flat.data_ptr() + 4 * o for p, o in zip(params, off))
    assert torch.equal(bucket.flat[off[1]:off[2]], torch.full((params[1].numel(),), 7.0))
    assert torch.equal(bucket.flat[off[2]:off[3]], torch.zeros(params[2].numel()))
    assert torch.equal(bucket.flat[off[0]:off[1]], before[off[0]:off[1]]) and torch.equal(bucket.flat[off[3]:off[4]], before[off[3]:off[4]])
    assert params[0].grad is views[0] and params[3].grad is views[3]


class _TinyMIL(torch.nn.Module):
    """[1, N, D] -> logits [1, C] (mean pooling + two linears): a stand-in single-head model for the host-side loop tests."""

    def __init__(self, d=6, c=3):
        super().__init__()
        self.a, self.b = torch.nn.Linear(d, 5), torch.nn.Linear(5, c)

    def forward(self, x):
        return self.b(torch.tanh(self.a(x[0].float())).mean(0, keepdim=True))


def _group_epoch_worker(rank, world, port, out, bags_per_step):
    """train.train_one_epoch with conf.bags_per_step > 1 (train_one_epoch_groups): staged groups, mean gradient per group, ONE bucket
    all-reduce per group."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = _TinyMIL()
    T.broadcast_parameters(model, world)
    conf = T.Struct(wd=1e-2, lr=1e-2, min_lr=0.0, warmup_epoch=0, train_epoch=4, seed=3, n_class=3, n_token=1, arch="abmil",
                    bags_per_step=bags_per_step)
    g = torch.Generator().manual_seed(5)
    data = [{"input": torch.randn(4 + i % 5, 6, generator=g).half(), "label": i % 3} for i in range(16)]
    bucket = T.GradBucket(list(model.parameters())) if world > 1 else None
    opt = T.make_optimizer(model, conf, torch.device("cpu"), bucket, lr=conf.lr)
    for epoch in range(2):
        stats = T.train_one_epoch(model, data, opt, torch.device("cpu"), epoch, conf, bucket, rank, world, log_every=0, fused=False)
        assert np.isfinite(stats["slide_loss"])
    if rank == 0:
        torch.save([p.detach().clone() for p in model.parameters()], out)
    if world > 1:
        dist.destroy_process_group()


def test_grouped_epoch_two_ranks_of_two_bags_equal_one_process_of_four():
    """bags_per_step under data parallelism: 2 gloo ranks x 2 slides per step == one process x 4 slides per step (same slides per
    optimizer step, mean gradient, same schedule), and a group step differs from four B = 1 steps."""
    with tempfile.TemporaryDirectory() as d:
        o2, o1, o0 = os.path.join(d, "w2.pt"), os.path.join(d, "w1.pt"), os.path.join(d, "w0.pt")
        mp.spawn(_group_epoch_worker, args=(2, _free_port(), o2, 2), nprocs=2, join=True)
        _group_epoch_worker(0, 1, _free_port(), o1, 4)
        _group_epoch_worker(0, 1, _free_port(), o0, 1)
        a, b, c = (torch.load(o, weights_only=False) for o in (o2, o1, o0))
    for pa, pb in zip(a, b):
        assert torch.allclose(pa, pb, atol=1e-6, rtol=1e-5)
    assert any(not torch.allclose(pb, pc, atol=1e-4) for pb, pc in zip(b, c))


def test_train_group_prefetcher_cpu_concatenates_in_order():
    from acmil_amd.staging import staged_train_groups
    data = [{"input": torch.full((3 + i, 4), float(i)).half(), "label": i % 2} for i in range(7)]
    groups = list(staged_train_groups(data, [6, 0, 3, 2, 5], "cpu", 2))
    assert [g["indices"] for g in groups] == [[6, 0], [3, 2], [5]]
    assert [g["rows"] for g in groups] == [[9, 3], [6, 5], [8]]
    assert groups[0]["input"].shape == (12, 4) and float(groups[0]["input"][0, 0]) == 6.0 and float(groups[0]["input"][9, 0]) == 0.0
    assert groups[1]["labels"] == [1, 0]


def test_bench_secondary_lines_route_and_never_start_the_direct_leg(monkeypatch):
    """The driver's default line nests one entry per SECONDARY key: GA shapes go to ga_workload, the composed GigaPath family (groups of
    16 slides) to wide_workload, TransMIL / train to other_workloads -- all with the side legs off (no CPU baseline, no per-slide loop, and
    never the opt-in direct-reduction child processes: the driver's multi-GPU run must not be able to wait on them); an entry that
    raises costs that entry only."""
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import argparse
    import bench
    seen = []

    def fake(name):
        def f(a, ctx):
            seen.append((name, a.workload, a.no_cpu_baseline, a.no_b1, getattr(a, "direct_reduce", None), a.batch, getattr(a, "bags_per_step", 1)))
            if a.workload == "transmil":
                raise RuntimeError("boom")
            return {"value": 1.0}
        return f
    monkeypatch.setattr(bench, "ga_workload", fake("ga"))
    monkeypatch.setattr(bench, "wide_workload", fake("wide"))
    monkeypatch.setattr(bench, "other_workloads", fake("other"))
    args = argparse.Namespace(workload="ga_eval", steps=20, warmup=5, batch=64, precision="f16x3", no_cpu_baseline=False, no_b1=False,
                              direct_reduce=True, bags_per_step=1, train_n=10000, gpus=1)
    out = bench.secondary_lines(args, (1, 0, torch.device("cpu")))
    assert set(out) == {k for k, _ in bench.SECONDARY}
    by = {s[1] + ("_g%d" % s[6] if s[6] > 1 else ""): s for s in seen}
    assert by["ga_cfg3"][0] == "ga" and by["ga_gigapath"][0] == "wide" and by["ga_gigapath"][5] == 16
    assert by["transmil"][0] == "other" and "error" in out["transmil"] and out["train_n10k"]["value"] == 1.0
    assert all(s[2] and s[3] and s[4] is False for s in seen)
    assert args.direct_reduce is True and args.no_b1 is False          # the caller's arguments are left alone
