import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.gcn_layers import GraphConvolution
torch.autograd.set_multithreading_enabled(False)
dv = torch.device("cuda:0"); B, n, D = 1024, 32, 300; dt = torch.bfloat16
g = torch.Generator().manual_seed(0)
x = torch.randn(B, n, D, generator=g).to(dt).to(dv).requires_grad_(True)
adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n); adj = (adj / adj.sum(-1, keepdim=True)).to(dt).to(dv)
layers = [GraphConvolution(D, D).to(dt).to(dv) for _ in range(3)]
G = torch.randn(B, n, D, generator=g).to(dt).to(dv)
for _ in range(5):
    h = x
    for l in layers: h = l(h, adj)
    h.backward(G)
torch.cuda.synchronize()
