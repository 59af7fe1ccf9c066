"""GPU parity tests of the training path: exact-fp32 MFMA GEMM, and the full ACMIL training step
(HIP forward + HIP backward through the C ABI) against the reference's own train_one_epoch captures."""
import numpy as np
import pytest
import torch

from conftest import TRAIN_CASES, ab_environ, case_dims, load_golden

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("ta,tb", [(False, False), (False, True), (True, False), (True, True)])
@pytest.mark.parametrize("shape", [(128, 128, 32), (300, 70, 129), (5, 2, 1000), (256, 512, 20000), (1, 1, 1)])
def test_gemm_matches_fp64(shape, ta, tb):
    from acmil_amd import ops
    m, n, k = shape
    g = torch.Generator().manual_seed(m * 7 + n * 3 + k)
    a = torch.randn((k, m) if ta else (m, k), generator=g)
    b = torch.randn((n, k) if tb else (k, n), generator=g)
    bias = torch.randn(n, generator=g)
    ref = (a.double().T if ta else a.double()) @ (b.double().T if tb else b.double()) * 0.5 + bias.double()
    out = ops.gemm(a.cuda(), b.cuda(), trans_a=ta, trans_b=tb, alpha=0.5, bias=bias.cuda())
    err = (out.cpu().double() - ref).abs().max().item()
    scale = (a.abs().double().T if ta else a.abs().double()) @ (b.abs().double().T if tb else b.abs().double())
    assert err <= 2e-6 * scale.max().item() + 1e-6, err


@pytest.mark.parametrize("shape", [(300, 200, 96), (1000, 384, 768), (77, 130, 50), (4096, 1152, 384), (1, 7, 384)])
def test_gemm_f16x3_matches_fp64(shape):
    """Split-f16 products (x @ W^T layout): ~1e-6 relative to |a|.|b|, with bias / relu / beta, ragged edges, fp16 B, split-K."""
    from acmil_amd import ops
    m, n, k = shape
    g = torch.Generator().manual_seed(m + 5 * n + k)
    a = torch.randn(m, k, generator=g) * 3.0
    b = torch.randn(n, k, generator=g) * 0.05
    bias = torch.randn(n, generator=g)
    ref = a.double() @ b.double().T + bias.double()
    scale = (a.abs().double() @ b.abs().double().T).max().item()
    out = ops.gemm(a.cuda(), b.cuda(), trans_b=True, bias=bias.cuda(), precision="f16x3")
    assert (out.cpu().double() - ref).abs().max().item() <= 3e-6 * scale + 1e-6
    out = ops.gemm(a.cuda(), b.cuda(), trans_b=True, bias=bias.cuda(), act=1, precision="f16x3")
    assert (out.cpu().double() - ref.clamp_min(0)).abs().max().item() <= 3e-6 * scale + 1e-6
    c0 = torch.randn(m, n, generator=g)
    out = ops.gemm(a.cuda(), b.cuda(), trans_b=True, out=c0.cuda().clone(), beta=1.0, precision="f16x3")
    assert (out.cpu().double() - (ref - bias.double() + c0.double())).abs().max().item() <= 3e-6 * scale + 1e-6
    bh = b.half()
    out = ops.gemm(a.cuda(), bh.cuda(), trans_b=True, precision="f16x3")
    assert (out.cpu().double() - a.double() @ bh.double().T).abs().max().item() <= 3e-6 * scale + 1e-6
    # the other three operand layouts ([K][rows] staging path)
    ref0 = a.double() @ b.double().T
    out = ops.gemm(a.cuda(), b.T.contiguous().cuda(), precision="f16x3")
    assert (out.cpu().double() - ref0).abs().max().item() <= 3e-6 * scale + 1e-6
    out = ops.gemm(a.T.contiguous().cuda(), b.cuda(), trans_a=True, trans_b=True, precision="f16x3")
    assert (out.cpu().double() - ref0).abs().max().item() <= 3e-6 * scale + 1e-6
    out = ops.gemm(a.T.contiguous().cuda(), b.T.contiguous().half().cuda(), trans_a=True, precision="f16x3")
    assert (out.cpu().double() - a.double() @ b.half().double().T).abs().max().item() <= 3e-6 * scale + 1e-6


def test_gemm_bf16x3_tiny_operands():
    """bf16 halves keep the fp32 exponent range: gradient-sized operands (1e-8) that f16 halves would flush."""
    from acmil_amd import ops
    g = torch.Generator().manual_seed(17)
    a = torch.randn(500, 256, generator=g) * 1e-8            # dS-like
    b = torch.randn(256, 300, generator=g) * 0.05            # weights, stored [K][N]
    ref = a.double() @ b.double()
    scale = (a.abs().double() @ b.abs().double()).max().item()
    out = ops.gemm(a.cuda(), b.cuda(), precision="bf16x3")
    assert (out.cpu().double() - ref).abs().max().item() <= 4e-5 * scale
    out = ops.gemm(a.T.contiguous().cuda(), b.cuda(), trans_a=True, precision="bf16x3")
    assert (out.cpu().double() - ref).abs().max().item() <= 4e-5 * scale
    rel = ((out.cpu().double() - ref).abs().sum() / ref.abs().sum()).item()
    assert rel < 1e-5, rel


def test_gemm_f16x3_tall_k_split():
    from acmil_amd import ops
    g = torch.Generator().manual_seed(11)
    a, b = torch.randn(64, 20000, generator=g), torch.randn(48, 20000, generator=g)
    out = ops.gemm(a.cuda(), b.cuda(), trans_b=True, precision="f16x3")
    ref = a.double() @ b.double().T
    tol = 3e-6 * (a.abs().double() @ b.abs().double().T).max().item()
    assert (out.cpu().double() - ref).abs().max().item() <= tol
    out = ops.gemm(a.T.contiguous().cuda(), b.T.contiguous().cuda(), trans_a=True, precision="f16x3")   # dW = dS^T h layout
    assert (out.cpu().double() - ref).abs().max().item() <= tol


def test_gemm_options():
    from acmil_amd import ops
    g = torch.Generator().manual_seed(3)
    a, b = torch.randn(200, 96, generator=g), torch.randn(96, 150, generator=g)
    c0 = torch.randn(200, 150, generator=g)
    aux = torch.randn(200, 150, generator=g)
    ref = a.double() @ b.double()
    out = ops.gemm(a.cuda(), b.cuda(), act=1)
    assert (out.cpu().double() - ref.clamp_min(0)).abs().max() < 1e-4
    out = ops.gemm(a.cuda(), b.cuda(), out=c0.cuda().clone(), beta=1.0)
    assert (out.cpu().double() - (ref + c0.double())).abs().max() < 1e-4
    out = ops.gemm(a.cuda(), b.cuda(), act=2, aux=aux.cuda())
    assert (out.cpu().double() - ref * (aux > 0).double()).abs().max() < 1e-4
    for dt in (torch.float16, torch.bfloat16):
        bh = b.to(dt)
        out = ops.gemm(a.cuda(), bh.cuda())
        assert (out.cpu().double() - a.double() @ bh.double()).abs().max() < 1e-4
    # batched, strided views (leading dimensions honoured)
    a3, b3 = torch.randn(4, 70, 48, generator=g), torch.randn(4, 48, 33, generator=g)
    out = ops.gemm(a3.cuda(), b3.cuda())
    assert (out.cpu().double() - a3.double() @ b3.double()).abs().max() < 1e-4
    big = torch.randn(64, 256, generator=g).cuda()
    out = ops.gemm(big[:, :128], big[:, 128:], trans_b=True)
    assert (out.cpu().double() - big[:, :128].cpu().double() @ big[:, 128:].cpu().double().T).abs().max() < 1e-4


def _losses(sub, slide, attn, label, k):
    """Step3_WSI_classification_ACMIL.py:201-216 on device tensors (trainer-side math)."""
    import torch.nn.functional as F
    loss0 = F.cross_entropy(sub, label.repeat_interleave(k)) if k > 1 else torch.zeros((), device=sub.device)
    loss1 = F.cross_entropy(slide, label)
    p = torch.softmax(attn, dim=-1)
    diff = torch.zeros((), device=sub.device)
    for i in range(k):
        for j in range(i + 1, k):
            diff = diff + torch.cosine_similarity(p[:, i], p[:, j], dim=-1).mean() / (k * (k - 1) / 2)
    return loss0, loss1, diff


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
@pytest.mark.parametrize("name", TRAIN_CASES)
def test_train_step_gradients_match_reference(name, precision):
    from acmil_amd.architecture.transformer import ACMIL_GA
    case, sd = load_golden(name)
    d, di, k, c = case_dims(sd)

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    model = ACMIL_GA(Conf, n_token=k, n_masked_patch=10, mask_drop=0.6, precision=precision)
    model.load_state_dict(sd)
    model = model.cuda().train()
    x = torch.from_numpy(case["x"]).cuda()          # fp16 bag, as stored on disk
    label = torch.from_numpy(case["label"]).cuda()
    sub, slide, attn = model(x.float(), uniforms=torch.from_numpy(case["uniforms"]).cuda())
    loss0, loss1, diff = _losses(sub, slide, attn, label, k)
    assert float(loss0.detach()) == pytest.approx(float(case["loss0"]), abs=2e-5)
    assert float(loss1.detach()) == pytest.approx(float(case["loss1"]), abs=2e-5)
    (diff + loss0 + loss1).backward()
    worst = 0.0
    for name_p, p in model.named_parameters():
        ref = case["grad." + name_p]
        assert p.grad is not None, name_p
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        scale = np.abs(ref).max() + 1e-12
        worst = max(worst, err / scale)
        assert err <= 2e-4 * scale + 1e-7, (name_p, err, scale)
    # one AdamW step with the reference's settings lands on the reference's post-step parameters
    opt = torch.optim.AdamW(model.parameters(), lr=float(case["lr"]), weight_decay=float(case["wd"]))
    opt.step()
    # (AdamW's first step is lr * g / (|g| + eps): elements whose reference gradient is below 1e-6 -- e.g. the bias of
    # attention_weights, whose gradient is analytically ~0 under the softmax -- are ill-conditioned and excluded)
    for name_p, p in model.named_parameters():
        ok = np.abs(case["grad." + name_p]) >= 1e-6
        got, ref = p.detach().cpu().numpy(), case["after." + name_p]
        assert np.abs(got - ref)[ok].max(initial=0.0) <= 2e-6, name_p


def test_half_bag_and_no_mask_training_paths():
    """fp16 bag fed directly (in-kernel convert) and n_masked_patch=0 training: gradients agree with the fp32-bag run."""
    from acmil_amd.architecture.transformer import ACMIL_GA
    case, sd = load_golden("ga_train_n640_d512_k5_c2")

    class Conf:
        D_feat, D_inner, n_class, n_token = 512, 256, 2, 5

    grads = []
    for xdt in (torch.float32, torch.float16):
        model = ACMIL_GA(Conf, n_token=5, n_masked_patch=0, precision="fp32")
        model.load_state_dict(sd)
        model = model.cuda().train()
        x = torch.from_numpy(case["x"]).to(xdt).cuda()
        sub, slide, attn = model(x)
        (sub.sum() + 2 * slide.sum() + attn.square().mean()).backward()
        grads.append([p.grad.clone() for p in model.parameters()])
    for a, b in zip(*grads):
        assert (a - b).abs().max() <= 1e-5 * a.abs().max() + 1e-8


def test_abmil_training_gradients_match_oracle_autograd():
    from acmil_amd.architecture.transformer import ABMIL
    from oracle import ga_oracle as O
    case, sd = load_golden("abmil_eval_n1000_d512_c2")

    class Conf:
        D_feat, D_inner, n_class, n_token = 512, 256, 2, 1

    model = ABMIL(Conf, precision="fp32")
    model.load_state_dict(sd)
    model = model.cuda().train()
    x = torch.from_numpy(case["x"])
    y = torch.tensor([1])
    torch.nn.functional.cross_entropy(model(x.cuda()), y.cuda()).backward()
    sdg = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    torch.nn.functional.cross_entropy(O.abmil_forward(x, sdg), y).backward()
    for name_p, p in model.named_parameters():
        ref = sdg[name_p].grad
        assert (p.grad.cpu() - ref).abs().max() <= 2e-4 * ref.abs().max() + 1e-7, name_p


@pytest.mark.parametrize("name", ["ga_train_n640_d512_k5_c2", "ga_train_n2048_d512_k5_c7"])
def test_fused_train_step_equals_reference_step(name):
    """model.train_step (fused HIP loss + backward, no autograd) reproduces the reference's losses and gradients."""
    from acmil_amd.architecture.transformer import ACMIL_GA
    case, sd = load_golden(name)
    d, di, k, c = case_dims(sd)

    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k

    model = ACMIL_GA(Conf, n_token=k, n_masked_patch=10, mask_drop=0.6, precision="f16x3")
    model.load_state_dict(sd)
    model = model.cuda().train()
    x = torch.from_numpy(case["x"]).cuda().unsqueeze(0)[0:1]
    losses, out = model.train_step(torch.from_numpy(case["x"]).cuda(), torch.from_numpy(case["label"]).cuda(),
                                   uniforms=torch.from_numpy(case["uniforms"]).cuda())
    l = losses.cpu().numpy()
    assert l[0] == pytest.approx(float(case["loss0"]), abs=2e-5) and l[1] == pytest.approx(float(case["loss1"]), abs=2e-5)
    assert l[3] == pytest.approx(l[0] + l[1] + l[2], abs=1e-6)
    for name_p, p in model.named_parameters():
        ref = case["grad." + name_p]
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        assert err <= 2e-4 * np.abs(ref).max() + 1e-7, (name_p, err)


def test_fused_loss_matches_torch_autograd():
    from acmil_amd import ops
    g = torch.Generator().manual_seed(4)
    sub = torch.randn(5, 7, generator=g).cuda().requires_grad_(True)
    slide = torch.randn(1, 7, generator=g).cuda().requires_grad_(True)
    attn = (torch.randn(1, 5, 3000, generator=g) * 2).cuda()
    attn[0, 1, 17] = attn[0, 3, 99] = -1e9                      # masked positions
    attn.requires_grad_(True)
    y = torch.tensor([4]).cuda()
    l0, l1, df = _losses(sub, slide, attn, y, 5)
    (l0 + l1 + df).backward()
    losses, d_sub, d_slide, d_A = ops.ga_loss(sub.detach(), slide.detach()[0], attn.detach()[0], y)
    ref = torch.stack([l0, l1, df, l0 + l1 + df]).detach()
    assert (losses - ref).abs().max() < 2e-6
    assert (d_sub - sub.grad).abs().max() < 1e-6 and (d_slide - slide.grad[0]).abs().max() < 1e-6
    assert (d_A - attn.grad[0]).abs().max() <= 1e-4 * attn.grad.abs().max() + 1e-10
    assert d_A[1, 17] == 0 and d_A[3, 99] == 0


@pytest.mark.parametrize("n,reps", [(130, 60), (3000, 60), (10000, 25)])
def test_one_call_step_is_bitwise_reproducible(n, reps):
    """The single-launch kernels of the training step hand data between workgroups inside a launch (arrival tickets,
    write-through publishes, self-resetting counters).  Every reduction is fixed-order and atomics only count arrivals, so the
    same step repeated must give bitwise identical losses, scores, indices and gradients -- a stale read or a lost reset shows
    up here (tools/stress_train_step.py is the long version)."""
    from acmil_amd import synthetic as S, train as T
    conf = T.Struct(train_epoch=50, warmup_epoch=0, wd=1e-5, lr=1e-4, min_lr=0, n_class=7, n_token=5, n_masked_patch=10,
                    mask_drop=0.6, arch="ga", precision="f16x3", seed=1, D_feat=512, D_inner=256)
    torch.manual_seed(0)
    model = T.build_model(conf).cuda().train()
    x = S.synthetic_bag(n, 512, slide_idx=2)[0].half().cuda().unsqueeze(0)
    y = torch.tensor([3]).cuda()
    u = torch.rand(5, 10, generator=torch.Generator().manual_seed(n)).cuda()
    ref = None
    for r in range(reps):
        losses, out = model.train_step(x, y, uniforms=u)
        cur = [losses.clone(), out["A_out"].clone(), out["sub_preds"].clone(), out["masked_idx"].clone()] + [p.grad.clone() for p in model.parameters()]
        if ref is None:
            ref = cur
        else:
            assert all(torch.equal(a, b) for a, b in zip(ref, cur)), "repeat %d differs" % r


@pytest.mark.parametrize("di,k", [(256, 5), (128, 5), (256, 1)])
def test_backward_tile_kernel_geometries_match_three_launch_path(di, k, tmp_path):
    """The one-kernel gate side of the backward (csrc/ga_bwd_tile.hip: 64- or 32-row tiles, two workgroups per CU, weight fragments
    global -> registers) against the three-launch path it replaces (ACMIL_GA_BWD_TILE=0: G GEMM, gate pass, dpre GEMM) -- the library
    reads the knobs once, so every variant runs in its own process.  Ragged bag sizes (a tile with one row, rows past the bag, a bag
    smaller than a tile), both tile heights, the 128-wide family and the single-branch (n_token = 1) instance."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import sys, torch; sys.path.insert(0, %r)\n"
        "from acmil_amd import synthetic as S\n"
        "from acmil_amd.architecture.transformer import ACMIL_GA\n"
        "di, k = int(sys.argv[2]), int(sys.argv[3])\n"
        "class Conf: D_feat, D_inner, n_class, n_token = 2 * di, di, 3, k\n"
        "torch.manual_seed(3)\n"
        "m = ACMIL_GA(Conf, n_token=k, n_masked_patch=10, mask_drop=0.6).cuda().train()\n"
        "out = []\n"
        "for i, n in enumerate([17, 33, 100, 1000, 4097, 21000]):\n"
        "    x = S.synthetic_bag(n, 2 * di, slide_idx=40 + i)[0].half().cuda().unsqueeze(0)\n"
        "    u = torch.rand(k, min(10, n), generator=torch.Generator().manual_seed(n)).cuda()\n"
        "    losses, _ = m.train_step(x, torch.tensor([i %% 3], device='cuda'), uniforms=u)\n"
        "    out.append((losses.cpu(), [p.grad.clone().cpu() for p in m.parameters()]))\n"
        "torch.save(out, sys.argv[1])\n" % root)
    res = {}
    for tag, env in (("three", {"ACMIL_GA_BWD_TILE": "0"}), ("auto", {}), ("r32", {"ACMIL_GA_BWD_ROWS": "32"}), ("r64", {"ACMIL_GA_BWD_ROWS": "64"})):
        e = ab_environ(**env)
        path = str(tmp_path / (tag + ".pt"))
        r = subprocess.run([sys.executable, "-c", code, path, str(di), str(k)], env=e, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-2000:]
        res[tag] = torch.load(path)
    for tag in ("auto", "r32", "r64"):
        for (l0, g0), (l1, g1) in zip(res["three"], res[tag]):
            assert torch.equal(l0, l1), tag                       # the forward and the losses do not depend on the backward's geometry
            for a, b in zip(g0, g1):
                # other split / summation order only; absolute floor: d bw = sum_n dA is a cancellation (softmax gradients sum to zero)
                assert (a - b).abs().max().item() <= 2e-5 * a.abs().max().item() + 1e-7, tag
    for (l0, g0), (l1, g1) in zip(res["r32"], res["r64"]):        # (and deterministic: the two tile heights are each reproducible)
        assert all(torch.isfinite(a).all() for a in g0)
