#!/usr/bin/env python3
"""Stage-by-stage check of the attention backward's edge pass (k_gat_atp_bwd + k_gat_atp_src) against the oracle's closed-form
intermediates: g_sigma [E,H] (CSR-slot order), the direct part gxd, g_x, g_edge_embed.  Runs on the GPU box.
    python3 tests/debug_k2_stages.py   (a checker: it uses the oracle, so it lives under tests/)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
from oracle import recon_oracle as O
from recon_amd import gat_layers
from recon_amd.gat_layers import gat_heads
from recon_amd.graph import prepare_graph

cap = {}
orig = gat_layers._carve


def carve(dev, sizes):
    ws, ptrs = orig(dev, sizes)
    if len(sizes) == 11:
        cap["ws"], cap["ptrs"], cap["sizes"] = ws, ptrs, sizes
    return ws, ptrs


gat_layers._carve = carve
gat_layers._GAT_PATH = "atp"
d = torch.device("cuda:0")
for (B, n, e, F_, R, D, H) in ((32, 8, 56, 50, 50, 50, 1), (8, 16, 64, 200, 200, 200, 8), (6, 10, 40, 24, 16, 32, 3)):
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, R, seed=1)
    N, E = B * n, B * e
    g = torch.Generator().manual_seed(2)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    G = torch.randn(N, H * D, generator=g)
    graph = prepare_graph(edge.to(d), None, N)
    xd, eed, ad, a2d = (t.to(d).requires_grad_(True) for t in (x, ee, a, a2))
    out = gat_heads(xd, eed, ad, a2d, graph, None, 0.2, True)
    out.backward(G.to(d))
    torch.cuda.synchronize()
    refs = [O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, True, G[:, h * D:(h + 1) * D].double()) for h in range(H)]
    eid = graph.eid_long.cpu()
    gsig_ref = torch.stack([r["g_sigma"] for r in refs], 1)[eid]                       # [E,H] slot order
    g_x_ref = sum(r["g_x"] for r in refs)
    g_ee_ref = sum(r["g_edge_embed"] for r in refs)
    ws, ptrs = cap["ws"], cap["ptrs"]
    view = lambda p, nel: gat_layers._view_f32(ws, p, nel).cpu().double()
    gsig = view(ptrs[2], E * H).view(E, H)
    err = lambda a_, b_: float((a_ - b_).abs().max() / (b_.abs().max() + 1e-30))
    print("shape", (B, n, e, F_, R, D, H), "rel err: out %.2e  g_sigma %.2e  g_x %.2e  g_ee %.2e  g_a %.2e" % (
        err(out.detach().cpu().double(), torch.cat([r["out"] for r in refs], 1)), err(gsig, gsig_ref), err(xd.grad.cpu().double(), g_x_ref),
        err(eed.grad.cpu().double(), g_ee_ref), err(ad.grad.cpu().double(), torch.stack([r["g_a"] for r in refs]))))
    if err(gsig, gsig_ref) > 1e-3:
        bad = ((gsig - gsig_ref).abs() > 1e-3 * gsig_ref.abs().max()).nonzero()
        print("  g_sigma bad entries:", bad[:10].tolist(), "of", bad.shape[0], "; dst of first:", graph.dst[bad[0, 0]].item() if bad.numel() else None)
    if err(xd.grad.cpu().double(), g_x_ref) > 1e-3:
        dd = (xd.grad.cpu().double() - g_x_ref).abs()
        rows = (dd.max(1).values > 1e-3 * g_x_ref.abs().max()).nonzero().flatten()
        cols = (dd.max(0).values > 1e-3 * g_x_ref.abs().max()).nonzero().flatten()
        print("  g_x bad rows", rows[:12].tolist(), "n", rows.numel(), "bad cols", cols[:12].tolist(), "n", cols.numel())
    if err(eed.grad.cpu().double(), g_ee_ref) > 1e-3:
        dd = (eed.grad.cpu().double() - g_ee_ref).abs()
        rows = (dd.max(1).values > 1e-3 * g_ee_ref.abs().max()).nonzero().flatten()
        cols = (dd.max(0).values > 1e-3 * g_ee_ref.abs().max()).nonzero().flatten()
        print("  g_ee bad rows", rows[:12].tolist(), "n", rows.numel(), "bad cols", cols[:12].tolist(), "n", cols.numel())
