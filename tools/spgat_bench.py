#!/usr/bin/env python3
"""Secondary bench: full SpGAT (H heads + out_att) forward + backward at cfg 2, D = 200 and D = 25."""
import json, os, sys
import torch
DENSE = "--dense" in sys.argv      # pass relation_embed[edge_type] materialised (the reference's call) instead of None (read in place)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd import synth as O
from recon_amd.models import SpGAT
from recon_amd.gat_layers import gather_rows

def run(D, B=512, n=16, e=64, F_=200, H=8, nrel=64, iters=10):
    dv = torch.device("cuda:0")
    N, E = B * n, B * e
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, F_, generator=g).to(dv).requires_grad_(True)
    rel = torch.randn(nrel, F_, generator=g).to(dv).requires_grad_(True)
    _, edge, _ = O.synthetic_batched_graph(B, n, e, 4, 4, seed=0)
    et = torch.randint(0, nrel, (E,), generator=g).to(dv)
    edge = edge.to(dv)
    torch.manual_seed(0)
    m = SpGAT(N, F_, D, F_, 0.0, 0.2, H).to(dv)
    G = torch.randn(N, H * D, generator=g).to(dv)
    nohop = torch.tensor([])
    def step():
        for p in m.parameters(): p.grad = None
        x.grad = None; rel.grad = None
        out, out_rel = m(None, x, rel, edge, et, gather_rows(rel, et) if DENSE else None, nohop, nohop)
        out.backward(G)
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(json.dumps({"workload": "SpGAT (8 heads + out_att) fwd+bwd, cfg 2", "D_per_head": D, "ms_per_step": ms, "edges_per_s": E / ms * 1e3}))

if __name__ == "__main__":
    for D in ([int(v) for v in sys.argv[1:] if v.isdigit()] or [200, 25]):
        run(D)
