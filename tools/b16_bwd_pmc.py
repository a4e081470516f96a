#!/usr/bin/env python3
"""Two forward + backward steps of the n = 32 bf16 block propagation (the batched GEMMs of the backward) for rocprofv3 --pmc runs:
python3 tools/b16_bwd_pmc.py [B]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.propagation import propagate_blocks, make_start_embedding, get_head_indices, get_tail_indices
torch.autograd.set_multithreading_enabled(False)
dv = torch.device("cuda:0")
n, d, L = 32, 8, 3
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
C, S, dd = n * (n - 1), 16 * n, 16
g = torch.Generator().manual_seed(0)
Ts = [(torch.relu(torch.randn(8, C, dd * dd, generator=g)) * 0.02).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1).requires_grad_(True) for _ in range(L)]
ident = torch.eye(dd, device=dv, dtype=torch.bfloat16).requires_grad_(True)
tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
h0 = (torch.randn(8, C, S, 1, generator=g) * tmpl).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1, 1).requires_grad_(True)
head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
G = torch.randn(8, C, dd * L, generator=g).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1)
for _ in range(2):
    for t in Ts + [ident, h0]:
        t.grad = None
    propagate_blocks(Ts, ident, n, h0, "relu", head, tail).backward(G)
torch.cuda.synchronize()
