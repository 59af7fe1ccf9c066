"""Debug: group step vs serial (per-bag single steps averaged) vs the fp64 oracle mean, per-parameter relative differences."""
import sys, torch
sys.path.insert(0, ".")
from oracle import ga_oracle as O
from acmil_amd.architecture.transformer import ACMIL_GA

def ga(sd, k, c, d, di, **kw):
    class Conf:
        D_feat, D_inner, n_class, n_token = d, di, c, k
    m = ACMIL_GA(Conf, n_token=k, **kw); m.load_state_dict(sd); return m.cuda()

rows = [int(a) for a in sys.argv[1].split(",")] if len(sys.argv) > 1 else [700, 1300]
D, Di, K, C = 512, 256, 5, 7
G = len(rows)
sd = O.default_state_dict(D, Di, C, K)
bags = [O.synthetic_bag(n, D, 100 + i)[0].half() for i, n in enumerate(rows)]
labels = torch.tensor([i % C for i in range(G)])
us = torch.rand(G, K, 10, generator=torch.Generator().manual_seed(9))
g64 = None
for b in range(G):
    sdg = {n: v.clone().double().requires_grad_(True) for n, v in sd.items()}
    ref = O.acmil_ga_forward(bags[b].double().unsqueeze(0), sdg, n_token=K, n_masked_patch=10, mask_drop=0.6, uniforms=us[b].double(), training=True)
    l0, l1, dl = O.acmil_losses(ref["sub_preds"], ref["slide_pred"], ref["A_out"], labels[b:b + 1], K)
    (l0 + l1 + dl).backward()
    g = {n: v.grad / G for n, v in sdg.items()}
    g64 = g if g64 is None else {n: g64[n] + g[n] for n in g}
bags = [b.cuda() for b in bags]; labels = labels.cuda(); us = us.cuda()
m = ga(sd, K, C, D, Di, n_masked_patch=10, mask_drop=0.6).train()
for rep in range(3):
    l, o = m.train_step_batch(bags, labels, uniforms=us)
    gg = {n: p.grad.clone() for n, p in m.named_parameters()}
    ls, os_ = m._train_step_group_serial(torch.cat(bags), rows, labels, us, m._all_params(), None, None)
    worst_g, worst_s, worst_gs = ("", 0), ("", 0), ("", 0)
    for n, p in m.named_parameters():
        sc = g64[n].abs().max().item()
        if sc < 1e-9: continue
        eg = (gg[n].cpu().double() - g64[n]).abs().max().item() / sc
        es = (p.grad.cpu().double() - g64[n]).abs().max().item() / sc
        egs = (gg[n] - p.grad).abs().max().item() / sc
        if eg > worst_g[1]: worst_g = (n, eg)
        if es > worst_s[1]: worst_s = (n, es)
        if egs > worst_gs[1]: worst_gs = (n, egs)
    print("rep %d  group-vs-fp64 %s %.2e | serial-vs-fp64 %s %.2e | group-vs-serial %s %.2e" % (rep, worst_g[0], worst_g[1], worst_s[0], worst_s[1], worst_gs[0], worst_gs[1]))
