"""Determinism of the float32 wide-state training step: B graphs = copies of a few; every copy's gradients must equal the first's bit for bit."""
import os, sys, torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from recon_amd.propagation import build_block_adjacency, propagate, propagate_blocks, make_start_embedding, get_head_indices, get_tail_indices
d_ = torch.device("cuda:0")
n, d, L = 32, 8, 3
C, S, dd = n * (n - 1), 16 * n, 16
B, copies_of = int(sys.argv[1]) if len(sys.argv) > 1 else 1024, 8
act = sys.argv[2] if len(sys.argv) > 2 else "relu"
g = torch.Generator().manual_seed(31)
Ts = [torch.relu(torch.randn(copies_of, C, dd * dd, generator=g)) * 0.02 for _ in range(L)]
ident = torch.eye(dd) + 0.02 * torch.randn(dd, dd, generator=g)
tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
h0 = torch.randn(copies_of, C, S, 1, generator=g) * tmpl
head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(d_); tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(d_)
Gr = torch.randn(copies_of, C, dd * L, generator=g)
reps = B // copies_of
for mode in ("dense", "blocks"):
    for it in range(3):
        Tb = [t.to(d_).repeat(reps, 1, 1).requires_grad_(True) for t in Ts]
        Ib = ident.to(d_).requires_grad_(True)
        hb = h0.to(d_).repeat(reps, 1, 1, 1).requires_grad_(True)
        Gb = Gr.to(d_).repeat(reps, 1, 1)
        if mode == "dense":
            out = propagate([build_block_adjacency(t, Ib, n) for t in Tb], hb, act, head, tail)
        else:
            out = propagate_blocks(Tb, Ib, n, hb, act, head, tail)
        (out * Gb).sum().backward()
        bad = 0
        for t in [out, hb.grad] + [t.grad for t in Tb]:
            v = t.reshape(reps, copies_of, -1)
            bad += int((v != v[:1]).flatten(1).any(1).sum())
        print(mode, "run", it, "copies that differ from the first:", bad)
        del Tb, hb, Gb, out
        torch.cuda.empty_cache()
