"""Random shapes through the float32 wide-state training paths (dense chain form and block mode) against the float64 oracle."""
import os, sys, random, torch
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
from oracle import recon_oracle as O
from recon_amd.propagation import propagate, propagate_blocks, build_block_adjacency, get_head_indices, get_tail_indices, make_start_embedding
d_ = torch.device("cuda:0")
rng = random.Random(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
ncase = int(sys.argv[2]) if len(sys.argv) > 2 else 20
worst = 0.0
for case in range(ncase):
    n = rng.randint(11, 32); L = rng.randint(1, 3); B = rng.choice([1, 2, 3, 5, 9]); act = rng.choice(["relu", "tanh", "linear"]); per_batch = rng.random() < 0.6
    mode = rng.choice(["dense", "blocks"])
    d = 8; Cn, S, dd = n * (n - 1), 16 * n, 16
    g = torch.Generator().manual_seed(case * 131 + n)
    Ts = [torch.relu(torch.randn(B, Cn, dd * dd, generator=g)) * (0.6 / n) for _ in range(L)]
    ident = torch.eye(dd) + 0.02 * torch.randn(dd, dd, generator=g)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(B, Cn, S, 1, generator=g) if per_batch else torch.randn(Cn, S, 1, generator=g)) * tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]); tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Gr = torch.randn(B, Cn, dd * L, generator=g)
    def run(device, dt, fn):
        Tl = [t.clone().to(device=device, dtype=dt).requires_grad_(True) for t in Ts]
        I = ident.clone().to(device=device, dtype=dt).requires_grad_(True)
        h = h0.clone().to(device=device, dtype=dt).requires_grad_(True)
        out = fn(Tl, I, h, head.to(device), tail.to(device)); (out * Gr.to(device=device, dtype=dt)).sum().backward()
        return [out.detach()] + [t.grad for t in Tl] + [I.grad, h.grad]
    ref = run("cpu", torch.float64, lambda Tl, I, h, hd, tl: O.propagate([O.build_block_adjacency(t, I, n) for t in Tl], h, act, hd, tl, as_gemm=True))
    if mode == "dense":
        got = run(d_, torch.float32, lambda Tl, I, h, hd, tl: propagate([build_block_adjacency(t, I, n) for t in Tl], h, act, hd, tl))
    else:
        got = run(d_, torch.float32, lambda Tl, I, h, hd, tl: propagate_blocks(Tl, I, n, h, act, hd, tl))
    rel = max(float((a.cpu().double() - b).abs().max() / (b.abs().max() + 1e-30)) for a, b in zip(got, ref))
    worst = max(worst, rel)
    print("case %2d n=%2d L=%d B=%d %-6s per_batch=%d %-6s max rel err (of tensor max) %.2e" % (case, n, L, B, act, per_batch, mode, rel), flush=True)
    assert rel < 2e-4, "mismatch"
print("worst", worst)
