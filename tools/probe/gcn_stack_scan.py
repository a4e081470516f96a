"""Fused GraphConvolution stack forward: time against the number of layers and graphs (is it per-workgroup latency or throughput?)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from tools.secondary import _time
from recon_amd.gcn_layers import GraphConvolution, gcn_stack
dv = torch.device("cuda:0")
n, D = 32, 300
for B in (1024, 512, 256, 2048, 4096):
    for L in (2, 3, 6):
        g = torch.Generator().manual_seed(0)
        x = torch.randn(B, n, D, generator=g).to(torch.bfloat16).to(dv)
        adj = (torch.rand(B, n, n, generator=g) < 0.2).to(torch.bfloat16).to(dv)
        torch.manual_seed(0)
        layers = [GraphConvolution(D, D).to(dv).to(torch.bfloat16) for _ in range(L)]
        with torch.no_grad():
            t = _time(lambda: gcn_stack(x, adj, layers), 30)
        print("B %5d L %d  %.1f us" % (B, L, t * 1e6), flush=True)
