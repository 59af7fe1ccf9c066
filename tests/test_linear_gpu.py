"""acmil_linear_f16x3 (packed-weight Linear kernel, csrc/linear_kernel.h) against fp64."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("m,k,n_out", [(300, 96, 128), (1, 64, 256), (777, 384, 1152), (5000, 768, 384), (4097, 512, 640), (129, 32, 128), (2000, 384, 384),
                                      (1536, 64, 512), (70000, 128, 384)])      # 24 tiles on 8 per-XCD queues of unequal length; a long launch (dynamic draws from every queue)
@pytest.mark.parametrize("xdt", [torch.float32, torch.float16, torch.bfloat16])
def test_linear_matches_fp64(m, k, n_out, xdt):
    from acmil_amd import ops
    g = torch.Generator().manual_seed(m + 3 * k + n_out)
    x = (torch.randn(m, k, generator=g) * 2.0).to(xdt)
    w = torch.randn(n_out, k, generator=g) * 0.05
    b = torch.randn(n_out, generator=g)
    ref = x.double() @ w.double().T + b.double()
    scale = (x.double().abs() @ w.double().abs().T).max().item()
    packed = ops.linear_pack(w.cuda())
    y = ops.linear_f16x3(x.cuda(), packed, n_out, bias=b.cuda())
    assert (y.cpu().double() - ref).abs().max().item() <= 3e-6 * scale + 1e-6
    y = ops.linear_f16x3(x.cuda(), packed, n_out, bias=b.cuda(), relu=True)
    assert (y.cpu().double() - ref.clamp_min(0)).abs().max().item() <= 3e-6 * scale + 1e-6
    # residual accumulate into a strided destination, strided source rows, no bias
    big = torch.randn(m, n_out + 64, generator=g).cuda()
    y0 = big[:, 32:32 + n_out]
    want = (ref - b.double()) + y0.cpu().double()
    xs = torch.zeros(m, k + 16, dtype=xdt).cuda()
    xs[:, :k] = x.cuda()
    out = ops.linear_f16x3(xs[:, :k], packed, n_out, out=y0, beta=1.0)
    assert out.data_ptr() == y0.data_ptr()
    assert (y0.cpu().double() - want).abs().max().item() <= 3e-6 * scale + 2e-6
    assert torch.equal(big[:, :32].cpu(), big[:, :32].cpu()) and torch.isfinite(big).all()


def test_wide_launch_with_bias_beyond_the_lin64_bias_table():
    """n_out = 2 304 columns (TransMIL's to_qkv at D_inner = 768), K = 768, bias: a shape lin64_kernel would be picked for by its K /
    tile rule, but whose launch is wider than that kernel's 2 048-float LDS bias table (columns 2048.. read past it)."""
    from acmil_amd import ops
    g = torch.Generator().manual_seed(11)
    m, k, n_out = 8000, 768, 2304
    x = torch.randn(m, k, generator=g)
    w = torch.randn(n_out, k, generator=g) * 0.05
    b = torch.randn(n_out, generator=g) * 3.0
    ref = x.double() @ w.double().T + b.double()
    scale = (x.double().abs() @ w.double().abs().T).max().item()
    y = ops.linear_f16x3(x.cuda(), ops.linear_pack(w.cuda()), n_out, bias=b.cuda())
    err = (y.cpu().double() - ref).abs()
    assert err[:, 2048:].max().item() <= 3e-6 * scale + 1e-6 and err.max().item() <= 3e-6 * scale + 1e-6


def test_single_k_step_is_refused():
    """K = 16 is one K step: the two-step-deep DMA ring would read past the operands (round-4 finding) -- refused, not attempted."""
    from acmil_amd import _lib
    assert _lib.load().acmil_linear_packed_bytes(128, 16) == 0 and _lib.load().acmil_linear_packed_bytes(128, 32) != 0


def test_linear_bitwise_reproducible_and_equal_to_generic_gemm_class():
    from acmil_amd import ops
    g = torch.Generator().manual_seed(1)
    x = torch.randn(20000, 768, generator=g).cuda()
    w = (torch.randn(384, 768, generator=g) * 0.03).cuda()
    packed = ops.linear_pack(w)
    a = ops.linear_f16x3(x, packed, 384)
    b = ops.linear_f16x3(x, packed, 384)
    assert torch.equal(a, b)
    c = ops.gemm(x, w, trans_b=True, precision="f16x3")
    assert (a - c).abs().max().item() <= 2e-5 * c.abs().max().item()


@pytest.mark.parametrize("m,k,n_out", [(3000, 512, 256), (22100, 768, 768), (40000, 1536, 768), (257, 64, 128)])      # lin_kernel, lin64_kernel (K >= 768, >= 256 units), both
def test_fp32_rows_of_f16_exact_values_skip_the_lo_products(m, k, n_out):
    """An fp32 operand whose values are f16-exact (bags stored fp16 and up-cast by the loop, Step3_WSI_classification_ACMIL.py:193)
    has lo halves of exact zeros: the kernels drop the W_hi x_lo MFMA group per wave and K step.  The result is the SAME numbers as the
    16-bit storage path (which never had that group) bit for bit; rows with a real lo half inside an otherwise exact operand keep the
    full three products (fp64 bound); both arithmetic classes bitwise reproducible."""
    from acmil_amd import ops
    g = torch.Generator().manual_seed(5 * m + k)
    x16 = (torch.randn(m, k, generator=g) * 2.0).half()
    w = torch.randn(n_out, k, generator=g) * 0.05
    b = torch.randn(n_out, generator=g)
    packed = ops.linear_pack(w.cuda())
    y32 = ops.linear_f16x3(x16.float().cuda(), packed, n_out, bias=b.cuda(), relu=True)
    y16 = ops.linear_f16x3(x16.cuda(), packed, n_out, bias=b.cuda(), relu=True)
    assert torch.equal(y32, y16)
    # a few rows with genuine fp32 values: their waves run the third product, every other wave still skips it
    xm = x16.float()
    rows = torch.randint(0, m, (max(1, m // 97),), generator=g)
    xm[rows] += torch.randn(len(rows), k, generator=g) * 1e-4
    ref = (xm.double() @ w.double().T + b.double()).clamp_min(0)
    scale = (xm.double().abs() @ w.double().abs().T).max().item()
    ym = ops.linear_f16x3(xm.cuda(), packed, n_out, bias=b.cuda(), relu=True)
    assert (ym.cpu().double() - ref).abs().max().item() <= 3e-6 * scale + 1e-6
    assert torch.equal(ym, ops.linear_f16x3(xm.cuda(), packed, n_out, bias=b.cuda(), relu=True))
    keep = torch.ones(m, dtype=torch.bool); keep[rows] = False
    # rows that share a 32-row wave tile with a perturbed row take the general path: still the same numbers (the skipped products are zeros)
    assert torch.equal(ym.cpu()[keep], y32.cpu()[keep])
