#!/usr/bin/env python3
"""One full-graph SpGAT step at the scale of the reference's stage-A data (FB15k-237-sized: 14 541 entities, 237 relations, 272 115
training triples, 100-d embeddings, 2 heads x 100 + out_att 200) with EVERY edge of the graph in one call and the same tensors every
step.  This is a size probe of the kernels (hub rows of ~9 000 edges, 272 k edges per launch), NOT the reference's training loop: that
one builds a fresh 128-entity batch per iteration (GAT/main.py:478-525) and is measured by tools/stage_a_iter_bench.py.
Synthetic stand-in for the data set (there is no network): head and tail entities drawn Zipf-like (exponent 0.8) so that the largest
entities have thousands of edges, as in the real graph.  fp32, forward and forward + backward."""
import json, os, sys
import numpy as np
import torch
DENSE = "--dense" in sys.argv      # pass relation_embed[edge_type] materialised (the reference's call) instead of None (read in place)
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.models import SpGAT
from recon_amd.gat_layers import gather_rows


def run(N=14541, E=272115, nrel=237, F_=100, D=100, H=2, nhop=0, iters=10):
    dv = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    p = 1.0 / np.arange(1, N + 1) ** 0.8; p /= p.sum()
    perm = rs.permutation(N)
    edge = torch.from_numpy(np.stack([perm[rs.choice(N, size=E, p=p)], rs.permutation(N)[rs.choice(N, size=E, p=p)]])).long().to(dv)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, F_, generator=g).to(dv).requires_grad_(True)
    rel = torch.randn(nrel, F_, generator=g).to(dv).requires_grad_(True)
    et = torch.randint(0, nrel, (E,), generator=g).to(dv)
    if nhop:
        edge_nhop = torch.from_numpy(np.stack([rs.choice(N, size=nhop, p=p), rs.randint(0, N, size=nhop)])).long().to(dv)
        et_nhop = torch.randint(0, nrel, (nhop, 2), generator=g).to(dv)
    else:
        edge_nhop = torch.tensor([]); et_nhop = torch.tensor([])
    torch.manual_seed(0)
    m = SpGAT(N, F_, D, F_, 0.0, 0.2, H).to(dv)
    G = torch.randn(N, H * D, generator=g).to(dv)
    deg = torch.bincount(edge[0], minlength=N); sdeg = torch.bincount(edge[1], minlength=N)

    def fwd():
        with torch.no_grad():
            m(None, x, rel, edge, et, rel[et] if DENSE else None, edge_nhop, et_nhop)

    def step():
        for q in m.parameters(): q.grad = None
        x.grad = None; rel.grad = None
        out, _ = m(None, x, rel, edge, et, gather_rows(rel, et) if DENSE else None, edge_nhop, et_nhop)
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
    Et = E + nhop
    print(json.dumps({"workload": "full-graph SpGAT step, FB15k-237-sized synthetic", "N": N, "E": Et, "nhop": nhop, "max_in_degree": int(deg.max()),
                      "max_out_degree": int(sdeg.max()), "heads": H, "D_per_head": D, **res, "edges_per_s_fwd_bwd": Et / res["fwd_bwd_ms"] * 1e3}))


def run_kbgat(N=14541, E=272115, nrel=237, nhop=100000, iters=10):
    """The reference's stage-A model as GAT/main.py builds it (embedding_size 50, entity_out_dim [100, 200], nheads_GAT [2, 2], drop 0.3,
    train mode): SpKBGATModified.forward on the whole graph + `nhop` 2-hop quadruples, and the backward of a scalar of its outputs."""
    from recon_amd.models import SpKBGATModified
    dv = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    p = 1.0 / np.arange(1, N + 1) ** 0.8; p /= p.sum()
    perm = rs.permutation(N)
    edge = torch.from_numpy(np.stack([perm[rs.choice(N, size=E, p=p)], rs.permutation(N)[rs.choice(N, size=E, p=p)]])).long().to(dv)
    g = torch.Generator().manual_seed(1)
    et = torch.randint(0, nrel, (E,), generator=g).to(dv)
    quads = torch.stack([torch.randint(0, N, (nhop,), generator=g), torch.randint(0, nrel, (nhop,), generator=g),
                         torch.randint(0, nrel, (nhop,), generator=g), torch.from_numpy(rs.choice(N, size=nhop, p=p))], dim=1).long().to(dv)
    batch_entities = torch.randint(0, N, (20000,), generator=g).to(dv)
    torch.manual_seed(0)
    m = SpKBGATModified(torch.randn(N, 50), torch.randn(nrel, 50), [100, 200], [100, 200], 0.3, 0.2, [2, 2]).to(dv)
    m.train()

    def step():
        m.zero_grad(set_to_none=True)
        ent, rel, _ = m(None, batch_entities, (edge, et), quads)
        (ent.sum() + rel.sum()).backward()
    for _ in range(3): step()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters): step()
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / iters
    print(json.dumps({"workload": "SpKBGATModified fwd+bwd (train mode, dropout 0.3), FB15k-237-sized synthetic", "N": N, "E": E + nhop, "nhop": nhop,
                      "fwd_bwd_ms": ms, "edges_per_s_fwd_bwd": (E + nhop) / ms * 1e3}))


if __name__ == "__main__":
    if "--kbgat" in sys.argv:
        run_kbgat()
    else:
        run()
        if "--nhop" in sys.argv: run(nhop=100000)
