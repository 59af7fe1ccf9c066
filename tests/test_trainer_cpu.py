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
