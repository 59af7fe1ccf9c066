"""TransMIL training path: every autograd Function against torch's CPU autograd of the same op, then the whole module's
gradients against the fixture captured from the real reference (dropout p = 0) and against the oracle's autograd."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from conftest import load_golden

pytestmark = pytest.mark.gpu


def _leaf(t):
    return t.clone().requires_grad_(True)


def _cmp(got, ref, tol, name=""):
    err = (got.detach().cpu().double() - ref.detach().double()).abs().max().item()
    scale = max(1.0, ref.detach().abs().max().item())
    assert err <= tol * scale, "%s: err %.3e (scale %.3e)" % (name, err, scale)


def test_layer_norm_fwd_bwd():
    from acmil_amd import autograd as AG
    g = torch.Generator().manual_seed(1)
    x, gam, bet, dy = torch.randn(1237, 384, generator=g) * 2 + 0.5, torch.randn(384, generator=g), torch.randn(384, generator=g), torch.randn(1237, 384, generator=g)
    xr, gr, br = _leaf(x), _leaf(gam), _leaf(bet)
    F.layer_norm(xr, (384,), gr, br, 1e-5).backward(dy)
    xg, gg, bg = _leaf(x.cuda()), _leaf(gam.cuda()), _leaf(bet.cuda())
    y = AG.layer_norm(xg, gg, bg, 1e-5)
    y.backward(dy.cuda())
    _cmp(y, F.layer_norm(x, (384,), gam, bet, 1e-5), 2e-6, "y")
    _cmp(xg.grad, xr.grad, 5e-6, "dx"); _cmp(gg.grad, gr.grad, 5e-6, "dgamma"); _cmp(bg.grad, br.grad, 5e-6, "dbeta")


@pytest.mark.parametrize("rows,cols", [(700, 192), (24, 5000), (3, 64)])
def test_softmax_rows_bwd(rows, cols):
    from acmil_amd import autograd as AG
    g = torch.Generator().manual_seed(2)
    s, dp = torch.randn(rows, cols, generator=g) * 3, torch.randn(rows, cols, generator=g)
    sr = _leaf(s); torch.softmax(sr, -1).backward(dp)
    sg = _leaf(s.cuda()); p = AG.softmax_rows(sg); p.backward(dp.cuda())
    _cmp(p, torch.softmax(s, -1), 1e-6, "p"); _cmp(sg.grad, sr.grad, 2e-6, "ds")


@pytest.mark.parametrize("precision", ["fp32", "f16x3"])
def test_matmul_and_linear_grads(precision):
    from acmil_amd import autograd as AG
    g = torch.Generator().manual_seed(3)
    tol = 2e-6 if precision == "fp32" else 6e-5
    x, w, b, dy = torch.randn(500, 96, generator=g), torch.randn(80, 96, generator=g) * 0.1, torch.randn(80, generator=g), torch.randn(500, 80, generator=g)
    xr, wr, br = _leaf(x), _leaf(w), _leaf(b)
    F.relu(F.linear(xr, wr, br)).backward(dy)
    xg, wg, bg = _leaf(x.cuda()), _leaf(w.cuda()), _leaf(b.cuda())
    y = AG.linear(xg, wg, bg, relu=True, precision=precision); y.backward(dy.cuda())
    _cmp(y, F.relu(F.linear(x, w, b)), tol, "y")
    _cmp(xg.grad, xr.grad, tol, "dx"); _cmp(wg.grad, wr.grad, tol * 30, "dw"); _cmp(bg.grad, br.grad, tol * 30, "db")
    # batched, strided head views as in the attention block: [h, n, d] views of a [n, 3*h*d] matrix
    n, h, d, m = 320, 8, 16, 64
    qkv, kl, dout = torch.randn(n, 3 * h * d, generator=g), torch.randn(h, m, d, generator=g), torch.randn(h, n, m, generator=g)
    qr, kr = _leaf(qkv), _leaf(kl)
    (qr[:, :h * d].reshape(n, h, d).permute(1, 0, 2) @ kr.transpose(1, 2) * 0.25).backward(dout)
    qg, kg = _leaf(qkv.cuda()), _leaf(kl.cuda())
    out = AG.matmul(qg[:, :h * d].reshape(n, h, d).permute(1, 0, 2), kg, trans_b=True, alpha=0.25, precision=precision)
    out.backward(dout.cuda())
    _cmp(qg.grad, qr.grad, tol * 10, "dqkv"); _cmp(kg.grad, kr.grad, tol * 30, "dkl")


def test_seq_conv_fwd_bwd():
    from acmil_amd import autograd as AG
    g = torch.Generator().manual_seed(4)
    n, di = 517, 128
    qkv, w, dout = torch.randn(n, 3 * di, generator=g), torch.randn(8, 1, 33, 1, generator=g) * 0.2, torch.randn(n, di, generator=g)
    qr, wr = _leaf(qkv), _leaf(w)
    v = qr[:, 2 * di:].reshape(1, n, 8, di // 8).permute(0, 2, 1, 3)                      # b h n d
    ref = F.conv2d(v, wr, padding=(16, 0), groups=8).permute(0, 2, 1, 3).reshape(n, di)
    ref.backward(dout)
    qg, wg = _leaf(qkv.cuda()), _leaf(w.cuda())
    out = AG.seq_conv(qg[:, 2 * di:], wg); out.backward(dout.cuda())
    _cmp(out, ref, 2e-6, "out"); _cmp(qg.grad, qr.grad, 5e-6, "dv"); _cmp(wg.grad, wr.grad, 2e-5, "dw")


def test_dwconv7_fwd_bwd():
    from acmil_amd import autograd as AG
    g = torch.Generator().manual_seed(5)
    side, c = 19, 128
    x, weff, beff, dy = torch.randn(side * side, c, generator=g), torch.randn(49, c, generator=g) * 0.1, torch.randn(c, generator=g), torch.randn(side * side, c, generator=g)
    xr, wr, br = _leaf(x), _leaf(weff), _leaf(beff)
    img = xr.t().reshape(1, c, side, side)
    ref = F.conv2d(img, wr.t().reshape(c, 1, 7, 7), br, padding=3, groups=c).flatten(2)[0].t()
    ref.backward(dy)
    xg, wg, bg = _leaf(x.cuda()), _leaf(weff.cuda()), _leaf(beff.cuda())
    y = AG.dwconv7(xg, wg, bg, side); y.backward(dy.cuda())
    _cmp(y, ref, 2e-6, "y"); _cmp(xg.grad, xr.grad, 5e-6, "dx"); _cmp(wg.grad, wr.grad, 2e-5, "dweff"); _cmp(bg.grad, br.grad, 2e-5, "dbeff")


def test_landmark_mean_fwd_bwd():
    from acmil_amd import autograd as AG
    g = torch.Generator().manual_seed(6)
    n, di, l = 64 * 7, 128, 7
    qkv, dout = torch.randn(n, 3 * di, generator=g), torch.randn(8, n // l, di // 8, generator=g)
    qr = _leaf(qkv)
    ref = qr[:, di:2 * di].reshape(n // l, l, 8, di // 8).mean(1).permute(1, 0, 2)
    ref.backward(dout)
    qg = _leaf(qkv.cuda())
    out = AG.landmark_mean(qg[:, di:2 * di], l); out.backward(dout.cuda())
    _cmp(out, ref, 1e-6, "out"); _cmp(qg.grad, qr.grad, 1e-6, "dsrc")


def _model(sd, d, di, c):
    from acmil_amd.architecture.transMIL import TransMIL

    class Conf:
        D_feat, D_inner, n_class = d, di, c
    m = TransMIL(Conf)
    m.load_state_dict(sd)
    return m.cuda()


def test_gradients_match_reference_golden():
    case, sd = load_golden("transmil_train_n300_d384_c2")
    model = _model(sd, 384, 128, 2).train()
    for layer in (model.layer1, model.layer2):
        layer.attn.to_out[1].p = 0.0                     # the fixture was captured with dropout disabled
    logits = model(torch.from_numpy(case["x"]).cuda())
    loss = F.cross_entropy(logits, torch.from_numpy(case["label"]).cuda())
    loss.backward()
    np.testing.assert_allclose(logits.detach().cpu().numpy(), case["logits"], rtol=0, atol=1e-4)
    assert abs(loss.item() - float(case["loss"])) < 1e-4
    worst = 0.0
    for name, p in model.named_parameters():
        ref = case["grad." + name]
        assert p.grad is not None, name
        err = np.abs(p.grad.cpu().numpy() - ref).max()
        worst = max(worst, err / max(1e-3, np.abs(ref).max()))
        assert err <= 2e-3 * max(1e-3, np.abs(ref).max()), "%s: %.3e vs max %.3e" % (name, err, np.abs(ref).max())
    assert worst < 2e-3


def test_gradients_match_oracle_autograd_other_shape():
    from oracle import transmil_oracle as TO
    d, di, c, n = 512, 256, 3, 777
    sd = TO.default_state_dict(d, di, c, seed=4)
    x = torch.randn(1, n, d, generator=torch.Generator().manual_seed(n))
    sdr = {k: v.clone().requires_grad_(True) for k, v in sd.items()}
    ref = TO.transmil_forward(x, sdr)["logits"]
    F.cross_entropy(ref, torch.tensor([2])).backward()
    model = _model(sd, d, di, c).eval()                 # eval mode + gradients enabled: same op-by-op path, dropout inactive
    logits = model(x.cuda())
    F.cross_entropy(logits, torch.tensor([2]).cuda()).backward()
    assert (logits.detach().cpu() - ref.detach()).abs().max() < 1e-4
    for name, p in model.named_parameters():
        r = sdr[name].grad
        err = (p.grad.cpu() - r).abs().max().item()
        assert err <= 3e-3 * max(1e-3, r.abs().max().item()), "%s: %.3e vs max %.3e" % (name, err, r.abs().max().item())


def test_train_mode_dropout_active_and_eval_equivalence():
    case, sd = load_golden("transmil_train_n300_d384_c2")
    model = _model(sd, 384, 128, 2)
    x = torch.from_numpy(case["x"]).cuda()
    model.train()
    a, b = model(x).detach(), model(x).detach()
    assert (a - b).abs().max() > 0                       # Dropout(0.1) draws differ between calls
    model.eval()
    with torch.no_grad():
        fused = model(x)                                  # fused eval pipeline
    opwise = model(x).detach()                            # op-by-op path (gradients enabled), dropout inactive in eval
    assert (fused - opwise).abs().max() < 1e-4
