"""Where do the block-mode and the materialised bf16 propagation differ?  (debug aid)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recon_amd.propagation import build_block_adjacency, propagate, propagate_blocks, make_start_embedding, get_head_indices, get_tail_indices
BF = torch.bfloat16
d_ = torch.device("cuda:0")
for n, L, B, act in [(9, 1, 2, "linear")]:
    d, dd, C, S = 8, 16, n * (n - 1), 16 * n
    g = torch.Generator().manual_seed(n + L)
    Ts = [(torch.relu(torch.randn(B, C, dd * dd, generator=g)) * (1.5 / S ** 0.5)).to(BF).to(d_) for _ in range(L)]
    I = (torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)).to(BF).to(d_)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(B, C, S, 1, generator=g) * tmpl).to(BF).to(d_)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(d_)
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(d_)
    with torch.no_grad():
        fused = propagate_blocks(Ts, I, n, h0, act, head, tail).float()
        plain = propagate([build_block_adjacency(t, I, n) for t in Ts], h0, act, head, tail).float()
    diff = (fused - plain).abs()
    bad = (diff > 0).nonzero()
    print("n=%d L=%d B=%d %s: %d of %d differ, max %.3e" % (n, L, B, act, bad.shape[0], diff.numel(), diff.max().item()))
    if bad.shape[0]:
        cs = sorted(set(bad[:, 1].tolist()))
        pairs = [(i, j) for i in range(n) for j in range(n) if i != j]
        print("   channels:", [(c, pairs[c]) for c in cs][:24])
        print("   columns:", sorted(set(bad[:, 2].tolist()))[:48])
        print("   graphs:", sorted(set(bad[:, 0].tolist())))
