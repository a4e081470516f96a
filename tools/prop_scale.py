import os, sys, torch
sys.path.insert(0, os.getcwd())
from recon_amd.propagation import build_block_adjacency, propagate, make_start_embedding, get_head_indices, get_tail_indices
dv = torch.device("cuda:0"); n, d, L = 9, 8, 3
C, S, dd = n*(n-1), 2*d*n, 2*d
for B in (128, 256, 512, 768, 1024, 2048):
    g = torch.Generator().manual_seed(0)
    adjs = [torch.randn(B, S, S, generator=g).to(dv) * 0.05 for _ in range(L)]
    h0 = torch.randn(B, C, S, 1, generator=g).to(dv)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv); tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
    with torch.no_grad():
        for _ in range(3): propagate(adjs, h0, "relu", head, tail)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): propagate(adjs, h0, "relu", head, tail)
        e1.record(); torch.cuda.synchronize()
    print("B=%5d  %.1f us" % (B, e0.elapsed_time(e1) * 1e3 / 20))
