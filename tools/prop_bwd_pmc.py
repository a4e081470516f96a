#!/usr/bin/env python3
"""Few forward+backward passes of the cfg-3b propagation for rocprofv3 --pmc runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.propagation import propagate, make_start_embedding, get_head_indices, get_tail_indices
dv = torch.device("cuda:0")
B, n, d, L = 1024, 9, 8, 3
C, S = n * (n - 1), 2 * d * n
g = torch.Generator().manual_seed(0)
adjs = [(torch.relu(torch.randn(B, S, S, generator=g)) * 0.05).to(dv).requires_grad_(True) for _ in range(L)]
tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
h0 = (torch.randn(B, C, S, 1, generator=g) * tmpl).to(dv).requires_grad_(True)
head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
G = torch.randn(B, C, 2 * d * L, generator=g).to(dv)
for _ in range(3):
    for t in adjs + [h0]:
        t.grad = None
    propagate(adjs, h0, "relu", head, tail).backward(G)
torch.cuda.synchronize()
