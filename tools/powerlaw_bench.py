#!/usr/bin/env python3
"""cfg-5-like workload (SURVEY 8d / BASELINE.md): power-law destination degrees, B graphs with n ~ U{16..256} nodes and
e = min(4096, 16 n) edges, dst ~ Zipf(1), src uniform; SpGAT (8 heads x D=25 + out_att), fp32, fwd and fwd+bwd."""
import json, os, sys
import numpy as np
import torch
DENSE = "--dense" in sys.argv      # pass relation_embed[edge_type] materialised (the reference's call) instead of None (read in place)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.models import SpGAT
from recon_amd.gat_layers import gather_rows

def batch(B, seed=0):
    rs = np.random.RandomState(seed)
    dsts, srcs, base = [], [], 0
    for _ in range(B):
        n = int(rs.randint(16, 257)); e = min(4096, 16 * n)
        p = 1.0 / np.arange(1, n + 1); p /= p.sum()
        dsts.append(rs.choice(n, size=e, p=p) + base); srcs.append(rs.randint(0, n, size=e) + base); base += n
    return torch.from_numpy(np.stack([np.concatenate(dsts), np.concatenate(srcs)])).long(), base

def run(B=64, D=25, F_=200, H=8, nrel=64, iters=10):
    dv = torch.device("cuda:0")
    edge, N = batch(B); E = edge.shape[1]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, F_, generator=g).to(dv).requires_grad_(True)
    rel = torch.randn(nrel, F_, generator=g).to(dv).requires_grad_(True)
    et = torch.randint(0, nrel, (E,), generator=g).to(dv); edge = edge.to(dv)
    torch.manual_seed(0)
    m = SpGAT(N, F_, D, F_, 0.0, 0.2, H).to(dv)
    G = torch.randn(N, H * D, generator=g).to(dv); nohop = torch.tensor([])
    deg = torch.bincount(edge[0], minlength=N)
    def fwd():
        with torch.no_grad():
            m(None, x, rel, edge, et, rel[et] if DENSE else None, nohop, nohop)
    def step():
        for p in m.parameters(): p.grad = None
        x.grad = None; rel.grad = None
        out, _ = m(None, x, rel, edge, et, gather_rows(rel, et) if DENSE else None, nohop, nohop)
        out.backward(G)
    res = {}
    for name, fn in (("fwd", fwd), ("fwd_bwd", step)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters): fn()
        e1.record(); torch.cuda.synchronize()
        res[name + "_ms"] = e0.elapsed_time(e1) / iters
    print(json.dumps({"workload": "power-law SpGAT (cfg 5-like)", "B": B, "N": N, "E": E, "max_degree": int(deg.max()), "D_per_head": D,
                      **res, "edges_per_s_fwd_bwd": E / res["fwd_bwd_ms"] * 1e3}))

if __name__ == "__main__":
    for B in ([int(v) for v in sys.argv[1:] if v.isdigit()] or [64, 512]):
        run(B)
