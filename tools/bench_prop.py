#!/usr/bin/env python3
"""Secondary benches (BASELINE.json configs[2], read as SURVEY 8d's cfg 3a / 3b):
  cfg 3b  GP-GNN block propagation  B=1024, n=9, 2d=16 (S=144, C=72), L=3, fp32
  cfg 3a  GraphConvolution stack    B=1024, n=32, D=300, 3 hops, fp32
Prints one JSON line per workload (fwd and fwd+bwd times from HIP events, block-edges/s)."""
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.propagation import (build_block_adjacency, propagate, make_start_embedding, get_head_indices,  # noqa: E402
                                   get_tail_indices)
from recon_amd.gcn_layers import GraphConvolution  # noqa: E402


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def prop(B=1024, n=9, d=8, L=3):
    dv = torch.device("cuda:0")
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(0)
    Ts = [(torch.relu(torch.randn(B, C, dd * dd, generator=g)) * 0.1).to(dv).requires_grad_(True) for _ in range(L)]
    ident = torch.eye(dd, device=dv, requires_grad=True)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(B, C, S, 1, generator=g) * tmpl).to(dv).requires_grad_(True)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
    G = torch.randn(B, C, dd * L, generator=g).to(dv)
    adjs = [build_block_adjacency(t, ident, n) for t in Ts]

    def fwd():
        with torch.no_grad():
            propagate([a.detach() for a in adjs], h0.detach(), "relu", head, tail)

    def fwd_bwd():
        for t in Ts + [ident, h0]:
            t.grad = None                                  # no accumulate kernels in the timed region
        a2 = [build_block_adjacency(t, ident, n) for t in Ts]
        out = propagate(a2, h0, "relu", head, tail)
        out.backward(G)
    tf, tb = timeit(fwd), timeit(fwd_bwd)
    flops = 2.0 * B * S * S * C * L
    bytes_alg = 4.0 * (L * B * S * S + B * C * S + B * C * dd * L)
    print(json.dumps({"workload": "cfg3b propagation fwd (adjacency prebuilt)", "B": B, "n": n, "S": S, "C": C, "L": L,
                      "fwd_us": tf * 1e6, "fwd_bwd_incl_adjacency_us": tb * 1e6,
                      "block_edges_per_s_fwd": B * n * n * L / tf, "fwd_TFLOPs": flops / tf / 1e12,
                      "fwd_GBps_algorithmic": bytes_alg / tf / 1e9}))


def gcn(B=1024, n=32, D=300, hops=3, dtype=torch.float32):
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    x = torch.randn(B, n, D, generator=g).to(dtype).to(dv).requires_grad_(True)
    adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
    adj = (adj / adj.sum(-1, keepdim=True)).to(dtype).to(dv)
    torch.manual_seed(0)
    layers = [GraphConvolution(D, D).to(dtype).to(dv) for _ in range(hops)]
    G = torch.randn(B, n, D, generator=g).to(dtype).to(dv)

    def fwd():
        with torch.no_grad():
            h = x
            for l in layers:
                h = l(h, adj)

    def fwd_bwd():
        h = x
        for l in layers:
            h = l(h, adj)
        h.backward(G)
    tf, tb = timeit(fwd), timeit(fwd_bwd)
    flops = hops * 2.0 * B * n * D * (D + n)
    for l in layers:
        l.weight.grad = None
        l.bias.grad = None
    print(json.dumps({"workload": "cfg3a GraphConvolution x%d fwd" % hops, "dtype": str(dtype).replace("torch.", ""), "B": B, "n": n, "D": D,
                      "fwd_us": tf * 1e6, "fwd_bwd_us": tb * 1e6, "dense_edges_per_s_fwd": B * n * n * hops / tf,
                      "fwd_TFLOPs": flops / tf / 1e12}))


if __name__ == "__main__":
    torch.autograd.set_multithreading_enabled(False)
    prop()
    gcn()
    gcn(dtype=torch.bfloat16)
    gcn(D=304, dtype=torch.bfloat16)
