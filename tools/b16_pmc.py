#!/usr/bin/env python3
"""Few launches of the n = 32 bf16 propagation forward (batched GEMM form) for rocprofv3 --pmc runs: python3 tools/b16_pmc.py [cfg]"""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.propagation import propagate, get_head_indices, get_tail_indices
if len(sys.argv) > 1:
    os.environ["RECON_BGEMM_CFG"] = sys.argv[1]
dv = torch.device("cuda:0")
n, d, L, B = 32, 8, 3, 1024
C, S = n * (n - 1), 16 * n
g = torch.Generator().manual_seed(0)
adjs = [((torch.rand(8, S, S, generator=g) - 0.45) * (2.0 / S ** 0.5)).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1) for _ in range(L)]
h0 = torch.randn(8, C, S, 1, generator=g).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1, 1)
head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
with torch.no_grad():
    for _ in range(2):
        propagate(adjs, h0, "relu", head, tail)
torch.cuda.synchronize()
