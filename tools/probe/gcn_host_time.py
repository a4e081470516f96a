"""Host time of one gcn_stack() training iteration at cfg 3a (no synchronisation inside the timed region; the queue is drained every 20 iterations)."""
import sys, time, cProfile, pstats
sys.path.insert(0, '/root/repo')
import torch
from recon_amd.gcn_layers import GraphConvolution, gcn_stack
torch.autograd.set_multithreading_enabled(False)
dv = torch.device("cuda:0"); dt = torch.bfloat16
B, n, D, hops = 1024, 32, 300, 3
g = torch.Generator().manual_seed(0)
x = torch.randn(B, n, D, generator=g).to(dt).to(dv).requires_grad_(True)
adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
adj = (adj / adj.sum(-1, keepdim=True)).to(dt).to(dv)
layers = [GraphConvolution(D, D).to(dt).to(dv) for _ in range(hops)]
G = torch.randn(B, n, D, generator=g).to(dt).to(dv)

def it():
    for l in layers:
        l.weight.grad = None; l.bias.grad = None
    x.grad = None
    t0 = time.perf_counter()
    y = gcn_stack(x, adj, layers)
    t1 = time.perf_counter()
    y.backward(G)
    t2 = time.perf_counter()
    return t1 - t0, t2 - t1
for _ in range(5): it()
torch.cuda.synchronize()
f = b = 0.0; N = 200
for i in range(N):
    a, c = it(); f += a; b += c
    if i % 20 == 19: torch.cuda.synchronize()
print("host us per iteration: forward %.1f backward %.1f" % (f / N * 1e6, b / N * 1e6))
pr = cProfile.Profile(); pr.enable()
for i in range(100):
    it()
    if i % 20 == 19: torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("tottime").print_stats(14)
