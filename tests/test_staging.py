"""Input staging: the prefetcher must hand out exactly dataset[i] in order, on CPU and through the
pinned-ring / copy-stream path on the GPU (ragged bags force ring regrowth and slot reuse)."""
import pytest
import torch

from acmil_amd.staging import BagPrefetcher


class _Bags:
    def __init__(self, sizes, d=64, dtype=torch.float16):
        g = torch.Generator().manual_seed(5)
        self.items = [{"input": torch.randn(n, d, generator=g).to(dtype), "label": i % 3} for i, n in enumerate(sizes)]

    def __len__(self):
        return len(self.items)

    def __getitem__(self, i):
        return self.items[i]


def test_prefetcher_cpu_order_and_content():
    data = _Bags([5, 1, 17, 9, 33, 2])
    order = [3, 0, 5, 1, 4, 2]
    got = list(BagPrefetcher(data, order, torch.device("cpu")))
    assert [g["index"] for g in got] == order
    for g in got:
        assert torch.equal(g["input"], data[g["index"]]["input"]) and g["label"] == data[g["index"]]["label"]


def test_prefetcher_propagates_reader_errors():
    class Bad(_Bags):
        def __getitem__(self, i):
            if i == 2:
                raise RuntimeError("corrupt bag")
            return super().__getitem__(i)
    with pytest.raises(RuntimeError, match="corrupt bag"):
        list(BagPrefetcher(Bad([4, 4, 4, 4]), [0, 1, 2, 3], torch.device("cpu")))


@pytest.mark.gpu
@pytest.mark.parametrize("depth", [2, 3, 5])
def test_prefetcher_gpu_ring_reuse(depth):
    sizes = [3000, 120, 9000, 9000, 1, 20000, 640, 20000, 77, 5000, 31000, 8, 31000, 2048]
    data = _Bags(sizes, d=128)
    order = list(range(len(sizes))) * 3
    dev = torch.device("cuda", 0)
    sums, refs = [], []
    for item in BagPrefetcher(data, order, dev, depth=depth):
        x = item["input"]
        assert x.is_cuda and x.dtype == torch.float16 and tuple(x.shape) == tuple(data[item["index"]]["input"].shape)
        # consume on the compute stream with a kernel that is slow relative to the copy (keeps slots busy)
        y = x.float()
        for _ in range(4):
            y = y * 1.0001 + 0.0
        sums.append((x.float().sum(), x.float().abs().max(), x[-1].float().sum()))
        refs.append(data[item["index"]]["input"])
    torch.cuda.synchronize()
    for (s, m, l), r in zip(sums, refs):
        rf = r.float().to(dev)
        assert torch.allclose(s, rf.sum(), rtol=1e-5, atol=1e-3)
        assert float(m) == float(rf.abs().max()) and torch.allclose(l, rf[-1].sum(), rtol=1e-5, atol=1e-3)


@pytest.mark.gpu
def test_staged_groups_respect_the_byte_budget():
    """the grouped eval ring is bounded by BYTES -- groups get shorter, every bag still arrives once, in order, intact."""
    from acmil_amd.staging import staged_groups
    sizes = [4000, 100, 12000, 12000, 50, 9000, 12000, 700, 12000, 3000, 12000, 12000, 64, 12000]
    data = _Bags(sizes, d=128)
    order = list(range(len(sizes))) * 2
    dev = torch.device("cuda", 0)
    budget = int(3.2 * 12000 * 128 * 2 * 1.25)          # about three of the largest bags (with the ring's 25 % head room)
    from acmil_amd.staging import BagPrefetcher
    pf = BagPrefetcher(data, order, dev, depth=16, max_bytes=budget)
    pf_groups = pf.iter_groups(8)
    seen, lens, checks = [], [], []
    for grp in pf_groups:
        lens.append(len(grp))
        for it in grp:
            seen.append(it["index"])
            checks.append((it["input"].float().sum(), data[it["index"]]["input"]))
    torch.cuda.synchronize()
    assert seen == order
    assert max(lens) < 8, "the budget, not the group size, must have closed the groups: %s" % lens
    # the budget also holds when recycled slots are regrown (small bags first, large ones later): never more than max_bytes in the ring
    assert 0 < pf.peak_bytes <= budget, (pf.peak_bytes, budget)
    for s, r in checks:
        assert torch.allclose(s, r.float().to(dev).sum(), rtol=1e-5, atol=1e-3)
    # unbounded ring of the same loop: full groups
    lens2 = [len(g) for g in staged_groups(data, order, dev, group=8, max_bytes=1 << 40)]
    assert lens2[:3] == [8, 8, 8]
