#!/usr/bin/env python3
"""Forward-only (inference) timing of the H-head attention stage at cfg 2 through both formulations."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd import synth as O
from recon_amd import gat_layers
from recon_amd.graph import prepare_graph

B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
d = torch.device("cuda:0")
x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
g = torch.Generator().manual_seed(0)
a = torch.stack([O.xavier_normal((D, 3 * F_), 1.414, g) for _ in range(H)]).to(d)
a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)]).to(d)
xd, eed = x.to(d), ee.to(d)
graph = prepare_graph(edge.to(d), None, B * n)
outs = {}
for path in ("proj", "atp"):
    gat_layers._GAT_PATH = path
    with torch.no_grad():
        for _ in range(3):
            o = gat_layers.gat_heads(xd, eed, a, a2, graph, None, 0.2, True)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            o = gat_layers.gat_heads(xd, eed, a, a2, graph, None, 0.2, True)
        e1.record(); torch.cuda.synchronize()
    outs[path] = o
    print("%s forward: %.1f us" % (path, e0.elapsed_time(e1) * 1e3 / 20))
print("max |atp - proj| = %.2e" % (outs["atp"] - outs["proj"]).abs().max().item())
