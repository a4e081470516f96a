import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from recon_amd.propagation import propagate_blocks, propagate, build_block_adjacency, get_head_indices, get_tail_indices, make_start_embedding
n, L, B, act = 11, 1, 300, "relu"
d = 8; Cn, S, dd = n * (n - 1), 16 * n, 16
g = torch.Generator().manual_seed(7 * n + L)
Ts = [torch.relu(torch.randn(B, Cn, dd * dd, generator=g)) * (0.6 / n) for _ in range(L)]
for t in Ts: t[:, ::5] *= 6.0
ident = torch.eye(dd) + 0.02 * torch.randn(dd, dd, generator=g)
tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
h0 = torch.randn(Cn, S, 1, generator=g) * tmpl
head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]); tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
Gr = torch.randn(B, Cn, dd * L, generator=g)
dv = torch.device("cuda:0")
def run(fn):
    Tl = [t.clone().to(dv).requires_grad_(True) for t in Ts]; I = ident.clone().to(dv).requires_grad_(True); h = h0.clone().to(dv).requires_grad_(True)
    out = fn(Tl, I, h, head.to(dv), tail.to(dv)); (out * Gr.to(dv)).sum().backward()
    return out.detach(), [t.grad for t in Tl], I.grad, h.grad
blk = run(lambda Tl, I, h, hd, tl: propagate_blocks(Tl, I, n, h, act, hd, tl))
dense = run(lambda Tl, I, h, hd, tl: propagate([build_block_adjacency(t, I, n) for t in Tl], h, act, hd, tl))
e = (blk[1][0] - dense[1][0]).abs()
print("g_T max", dense[1][0].abs().max().item(), "err max", e.max().item())
pg = e.view(B, -1).max(1).values
print("per-graph err (first 8, 250..262, last 4):", pg[:8].tolist(), pg[250:262].tolist(), pg[-4:].tolist())
b = int(pg.argmax()); eb = e[b].view(Cn, 16, 16)
pe = eb.view(Cn, -1).max(1).values
print("worst graph", b, "worst blocks", torch.topk(pe, 6))
c = int(pe.argmax()); print("block", c, "i,j =", c // (n - 1), c % (n - 1)); print(eb[c])
print("g_I err", (blk[2] - dense[2]).abs().max().item(), "g_h err", (blk[3] - dense[3]).abs().max().item(), "out err", (blk[0] - dense[0]).abs().max().item())
