"""Secondary workloads of BASELINE.json (configs[2] read as SURVEY 8d's cfg 3a / 3b, configs[4] as cfg 5), each timed with HIP events
and priced with SURVEY 8d's own byte / flop formulas.  bench.py puts these figures into its one JSON line (`secondary`); every
function returns a dict  {"ms": ..., "bytes" | "flops": ..., "bound": "hbm" | "mfma", "frac": ..., ...}  and frees what it allocated.

Peaks: HBM 8.0 TB/s, fp32 MFMA 157.3 TF/s, dense f16 MFMA 2.5 PF/s (/opt/skills/guides/MI355X_MICROARCH.md)."""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

HBM_PEAK = 8.0e12
MFMA_F32_PEAK = 157.3e12
MFMA_F16_PEAK = 2500e12


def _time(fn, iters, warm=2):
    """Seconds per call: the faster of two timed batches of `iters` calls (a fresh box pages code objects and grows the allocator's
    pools on first use; one such stall inside a batch of six calls would be the whole figure)."""
    for _ in range(warm):
        fn()
    best = None
    for _ in range(2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(iters):
            fn()
        e1.record()
        torch.cuda.synchronize()
        t = e0.elapsed_time(e1) / iters * 1e-3
        best = t if best is None or t < best else best
    return best


def propagation(n, B=1024, d=8, L=3, iters=10, with_backward=True):
    """cfg 3b: GP-GNN block form (models/models.py:240-274), 2d = 16, L = 3 untied, per-batch h0 from P4's template, relu.
    Forward (adjacency prebuilt, inference), forward with the states saved, and block adjacency + propagation forward + backward.
    Bytes: 4 (L B S^2 + B C S + B C 2d L); flops 2 B S^2 C L."""
    from recon_amd.propagation import (build_block_adjacency, propagate, propagate_blocks, blocks_mode_available, make_start_embedding,
                                       get_head_indices, get_tail_indices)
    dv = torch.device("cuda:0")
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(0)
    small = 8                                                        # n = 32: 1 GiB per T: build 8 graphs on the host, repeat on the device
    reps = B // small if n > 16 else 1
    Bh = small if n > 16 else B
    Ts = [(torch.relu(torch.randn(Bh, C, dd * dd, generator=g)) * (0.1 if n <= 16 else 0.02)).to(dv).repeat(reps, 1, 1).requires_grad_(True) for _ in range(L)]
    ident = torch.eye(dd, device=dv, requires_grad=True)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(Bh, C, S, 1, generator=g) * tmpl).to(dv).repeat(reps, 1, 1, 1).requires_grad_(True)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
    G = torch.randn(B, C, dd * L, generator=g).to(dv) if n <= 16 else torch.randn(Bh, C, dd * L, generator=g).to(dv).repeat(reps, 1, 1)
    with torch.no_grad():
        adjs = [build_block_adjacency(t, ident, n) for t in Ts]

    def fwd():
        with torch.no_grad():
            propagate(adjs, h0.detach(), "relu", head, tail)

    def fwd_bwd():
        for t in Ts + [ident, h0]:
            t.grad = None
        a2 = [build_block_adjacency(t, ident, n) for t in Ts]
        propagate(a2, h0, "relu", head, tail).backward(G)
    nbytes = 4.0 * (L * B * S * S + B * C * S + B * C * dd * L)
    flops = 2.0 * B * S * S * C * L
    tf = _time(fwd, iters)
    res = {"B": B, "n": n, "S": S, "C": C, "L": L, "fwd_ms": tf * 1e3, "bytes": nbytes, "flops": flops,
           "GBps": nbytes / tf / 1e9, "TFLOPs": flops / tf / 1e12}
    if S <= 160:     # two-term f16 kernel, whole graph per workgroup: 11 us of matrix work against one 39 us pass over the adjacency stack: HBM binds
        res.update(bound="hbm", frac=nbytes / tf / HBM_PEAK, kernel="k_propagate_fwd_h")
    else:            # wide states: two-term f16 kernel in 64-channel chunks (+ the split pass): 3 f16 MFMAs per fp32 product, priced as issued
        #              against the dense f16 peak; the 5.5 GB of algorithmic bytes would take 0.7 ms — the matrix pipe binds
        res.update(bound="mfma", mfma_flops_issued=3 * flops, peak_tflops=MFMA_F16_PEAK / 1e12, frac=3 * flops / tf / MFMA_F16_PEAK,
                   kernel="k_prop_split_adj + k_propagate_fwd_hl", hbm_frac=nbytes / tf / HBM_PEAK)
    if with_backward:
        tb = _time(fwd_bwd, max(2, iters // 2))
        res["fwd_bwd_incl_adjacency_ms"] = tb * 1e3
        # models/models.py:240-274 in one call: A_l read out of the transition tensors in place (inference: any n <= 32; with gradients, where the
        # backward's two-term form exists: d T written in T's layout)

        def fused():
            for t in Ts + [ident, h0]:
                t.grad = None
            propagate_blocks(Ts, ident, n, h0, "relu", head, tail).backward(G)

        def fused_fwd():
            with torch.no_grad():
                propagate_blocks(Ts, ident, n, h0, "relu", head, tail)

        def unfused_fwd_incl_adjacency():
            with torch.no_grad():
                propagate([build_block_adjacency(t, ident, n) for t in Ts], h0, "relu", head, tail)
        if blocks_mode_available(B, n, dd, h0, need_grad=False):
            res["fwd_incl_adjacency_ms"] = _time(unfused_fwd_incl_adjacency, iters) * 1e3
            res["fused_blocks_fwd_ms"] = _time(fused_fwd, iters) * 1e3
        from recon_amd.propagation import _blocks_wide_trainable
        if blocks_mode_available(B, n, dd, h0) or (n > 10 and _blocks_wide_trainable(B, n, dd, h0, L, head, tail)):
            res["fused_blocks_fwd_bwd_ms"] = _time(fused, max(2, iters // 2)) * 1e3
    del Ts, adjs, h0, G
    torch.cuda.empty_cache()
    return res


def propagation_bf16(n, B=1024, d=8, L=3, iters=10, with_backward=True):
    """cfg 3b in bfloat16 (BASELINE.json configs[2] as stated): the same step on bf16 tensors — bf16 storage, fp32 accumulation, one MFMA per
    product (csrc/prop_b16.hip).  Bytes: SURVEY 8d's formula at s = 2: 2 (L B S^2 + B C S + B C 2d L); flops 2 B S^2 C L."""
    from recon_amd.propagation import (build_block_adjacency, propagate, propagate_blocks, make_start_embedding, get_head_indices, get_tail_indices)
    dv = torch.device("cuda:0")
    bf = torch.bfloat16
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(0)
    small = 8
    reps = B // small if n > 16 else 1
    Bh = small if n > 16 else B
    Ts = [(torch.relu(torch.randn(Bh, C, dd * dd, generator=g)) * (0.1 if n <= 16 else 0.02)).to(bf).to(dv).repeat(reps, 1, 1).requires_grad_(True)
          for _ in range(L)]
    ident = torch.eye(dd, device=dv, dtype=bf, requires_grad=True)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(Bh, C, S, 1, generator=g) * tmpl).to(bf).to(dv).repeat(reps, 1, 1, 1).requires_grad_(True)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
    G = torch.randn(Bh, C, dd * L, generator=g).to(bf).to(dv).repeat(reps, 1, 1)
    with torch.no_grad():
        adjs = [build_block_adjacency(t, ident, n) for t in Ts]

    def fwd():
        with torch.no_grad():
            propagate(adjs, h0.detach(), "relu", head, tail)

    def fwd_blocks():
        with torch.no_grad():
            propagate_blocks(Ts, ident, n, h0, "relu", head, tail)

    def fwd_bwd():
        for t in Ts + [ident, h0]:
            t.grad = None
        propagate_blocks(Ts, ident, n, h0, "relu", head, tail).backward(G)
    nbytes = 2.0 * (L * B * S * S + B * C * S + B * C * dd * L)
    flops = 2.0 * B * S * S * C * L
    tf = _time(fwd, iters)
    res = {"B": B, "n": n, "S": S, "C": C, "L": L, "dtype": "bf16", "fwd_ms": tf * 1e3, "bytes": nbytes, "flops": flops,
           "GBps": nbytes / tf / 1e9, "TFLOPs": flops / tf / 1e12}
    if S <= 160:
        res.update(bound="hbm", frac=nbytes / tf / HBM_PEAK, kernel="k_prop_b16_fwd")
    else:
        res.update(bound="mfma", peak_tflops=MFMA_F16_PEAK / 1e12, frac=flops / tf / MFMA_F16_PEAK, kernel="k_bgemm_b16 x L + k_prop_b16_gather",
                   hbm_frac=nbytes / tf / HBM_PEAK)
    res["fwd_from_transition_tensors_ms"] = _time(fwd_blocks, iters) * 1e3
    if with_backward:
        res["fwd_bwd_incl_adjacency_ms"] = _time(fwd_bwd, max(2, iters // 2)) * 1e3
    del Ts, adjs, h0, G
    torch.cuda.empty_cache()
    return res


def gcn_bf16(B=1024, n=32, D=300, hops=3, iters=10):
    """cfg 3a: GraphConvolution x 3 (models/layers.py:57-63) in bf16 storage / fp32 accumulation.  Bytes (unfused, s = 2):
    per hop 2 B n D s + B n^2 s + D^2 s; flops per hop 2 B n D (D + n)."""
    from recon_amd.gcn_layers import GraphConvolution
    dv = torch.device("cuda:0")
    g = torch.Generator().manual_seed(0)
    dt = torch.bfloat16
    x = torch.randn(B, n, D, generator=g).to(dt).to(dv).requires_grad_(True)
    adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
    adj = (adj / adj.sum(-1, keepdim=True)).to(dt).to(dv)
    torch.manual_seed(0)
    layers = [GraphConvolution(D, D).to(dt).to(dv) for _ in range(hops)]
    G = torch.randn(B, n, D, generator=g).to(dt).to(dv)

    def fwd():
        with torch.no_grad():
            h = x
            for l in layers:
                h = l(h, adj)

    def fwd_bwd():
        for l in layers:
            l.weight.grad = None
            l.bias.grad = None
        x.grad = None
        h = x
        for l in layers:
            h = l(h, adj)
        h.backward(G)
    from recon_amd.gcn_layers import gcn_stack
    for l in layers:
        l.eval()

    def fwd_stack():
        with torch.no_grad():
            gcn_stack(x, adj, layers)
    def fwd_bwd_stack():
        for l in layers:
            l.weight.grad = None
            l.bias.grad = None
        x.grad = None
        gcn_stack(x, adj, layers).backward(G)
    s = 2.0
    nbytes = hops * (2 * B * n * D * s + B * n * n * s + D * D * s)
    nbytes_fused = 2 * B * n * D * s + B * n * n * s + hops * D * D * s          # H resident on chip across the hops
    flops = hops * 2.0 * B * n * D * (D + n)
    tf, tb, ts = _time(fwd, iters), _time(fwd_bwd, max(2, iters // 2)), _time(fwd_stack, iters)
    tsb = _time(fwd_bwd_stack, max(2, iters // 2))
    return {"B": B, "n": n, "D": D, "hops": hops, "dtype": "bf16", "fwd_ms": tf * 1e3, "fwd_bwd_ms": tb * 1e3, "bytes": nbytes, "flops": flops,
            "bound": "hbm", "frac": nbytes / tf / HBM_PEAK, "GBps": nbytes / tf / 1e9, "TFLOPs": flops / tf / 1e12,
            "dense_edges_per_s_fwd": B * n * n * hops / tf,
            "fused_stack": {"fwd_ms": ts * 1e3, "fwd_bwd_ms": tsb * 1e3, "bytes": nbytes_fused, "frac": nbytes_fused / ts / HBM_PEAK, "frac_of_bf16_mfma": flops / ts / MFMA_F16_PEAK,
                            "kernel": "k_gcn_b16_stack_fwd (gcn_stack(): all hops in one launch, activations in LDS)"}}


def powerlaw_spgat(B=64, D=25, F_=200, H=8, nrel=64, iters=10):
    """cfg 5-like: power-law destination degrees (n ~ U{16..256}, e = min(4096, 16 n), dst ~ Zipf(1)), full SpGAT (H heads + out_att),
    fp32, relation table read in place.  Bytes: SURVEY 8d's B_G for the two attention layers' forward edge stages
    (4 HD (E + 3 N) + 4 (E + N + 1) + 4 HD with HD = H D for the heads and for out_att)."""
    from recon_amd.models import SpGAT
    dv = torch.device("cuda:0")
    rs = np.random.RandomState(0)
    dsts, srcs, base = [], [], 0
    for _ in range(B):
        n = int(rs.randint(16, 257))
        e = min(4096, 16 * n)
        p = 1.0 / np.arange(1, n + 1)
        p /= p.sum()
        dsts.append(rs.choice(n, size=e, p=p) + base)
        srcs.append(rs.randint(0, n, size=e) + base)
        base += n
    edge = torch.from_numpy(np.stack([np.concatenate(dsts), np.concatenate(srcs)])).long().to(dv)
    N, E = base, edge.shape[1]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, F_, generator=g).to(dv).requires_grad_(True)
    rel = torch.randn(nrel, F_, generator=g).to(dv).requires_grad_(True)
    et = torch.randint(0, nrel, (E,), generator=g).to(dv)
    torch.manual_seed(0)
    m = SpGAT(N, F_, D, F_, 0.0, 0.2, H).to(dv)
    G = torch.randn(N, H * D, generator=g).to(dv)
    nohop = torch.tensor([])

    def fwd():
        with torch.no_grad():
            m(None, x, rel, edge, et, None, nohop, nohop)

    def fwd_bwd():
        for q in m.parameters():
            q.grad = None
        x.grad = None
        rel.grad = None
        out, _ = m(None, x, rel, edge, et, None, nohop, nohop)
        out.backward(G)
    HD = H * D
    b_g = 2 * (4.0 * HD * (E + 3 * N) + 4.0 * (E + N + 1) + 4.0 * HD)
    tf, tb = _time(fwd, iters), _time(fwd_bwd, max(2, iters // 2))
    deg = torch.bincount(edge[0], minlength=N)
    return {"graphs": B, "N": N, "E": E, "max_in_degree": int(deg.max()), "D_per_head": D, "heads": H, "fwd_ms": tf * 1e3, "fwd_bwd_ms": tb * 1e3,
            "edges_per_s_fwd_bwd": E / tb, "bytes": b_g, "bound": "hbm", "frac": b_g / tf / HBM_PEAK,
            "note": "frac prices SURVEY 8d's B_G of both layers against the WHOLE forward (GEMMs, row sums, hub combines included)"}


def powerlaw_mixed_stack_bf16(B=64, D=32, F_=200, H=8, iters=10):
    """BASELINE.json configs[4] as stated: power-law graphs of up to 256 nodes / 4 096 edges, a mixed stack in bfloat16 — an H-head attention
    layer (bf16 features in / out, GAT/layers.py:111-178) followed by three GraphConvolutions applied per graph (every graph its own dense
    row-normalised adjacency: a ragged batch, models/layers.py:57-63).  Bytes: SURVEY 8d's B_G for the attention layer's forward edge stage +
    per convolution 2 N D s + sum n_b^2 s + D^2 s at s = 2."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    from recon_amd.gcn_layers import GraphConvolution, RaggedAdjacency
    dv = torch.device("cuda:0")
    bf = torch.bfloat16
    rs = np.random.RandomState(0)
    dsts, srcs, mats, sizes, base = [], [], [], [], 0
    for _ in range(B):
        n = int(rs.randint(16, 257))
        e = min(4096, 16 * n)
        p = 1.0 / np.arange(1, n + 1)
        p /= p.sum()
        dl, sl = rs.choice(n, size=e, p=p), rs.randint(0, n, size=e)
        a = torch.zeros(n, n)
        a[torch.from_numpy(dl), torch.from_numpy(sl)] = 1.0
        a += torch.eye(n)
        mats.append((a / a.sum(-1, keepdim=True)).to(bf))
        dsts.append(dl + base); srcs.append(sl + base); sizes.append(n)
        base += n
    N = base
    edge = torch.from_numpy(np.stack([np.concatenate(dsts), np.concatenate(srcs)])).long().to(dv)
    E = edge.shape[1]
    g = torch.Generator().manual_seed(1)
    x = torch.randn(N, F_, generator=g).to(bf).to(dv).requires_grad_(True)
    ee = (torch.randn(E, F_, generator=g) * 0.5).to(bf).to(dv)
    HD = H * D
    a = (torch.randn(H, D, 3 * F_, generator=g) * (2.0 / (3 * F_ + D)) ** 0.5).to(dv).requires_grad_(True)
    a2 = (torch.randn(H, D, generator=g) * (2.0 / (D + 1)) ** 0.5).to(dv).requires_grad_(True)
    graph = prepare_graph(edge, None, N)
    rag = RaggedAdjacency.from_dense([m.to(dv) for m in mats])
    torch.manual_seed(0)
    layers = [GraphConvolution(HD, HD).to(bf).to(dv) for _ in range(3)]
    G = torch.randn(N, HD, generator=g).to(bf).to(dv)

    def stack():
        h = gat_layers.gat_heads(x, ee, a, a2, graph, None, 0.2, True)
        for l in layers:
            h = l(h, rag)
        return h

    def fwd():
        with torch.no_grad():
            stack()

    def fwd_bwd():
        for q in [x, a, a2] + [t for l in layers for t in l.parameters()]:
            q.grad = None
        stack().backward(G)
    b_g = 4.0 * HD * (E + 3 * N) + 4.0 * (E + N + 1) + 4.0 * HD
    b_c = 3 * (2.0 * N * HD * 2 + sum(v * v for v in sizes) * 2.0 + HD * HD * 2.0)
    tf, tb = _time(fwd, iters), _time(fwd_bwd, max(2, iters // 2))
    return {"graphs": B, "N": N, "E": E, "max_nodes": max(sizes), "heads": H, "D_per_head": D, "dtype": "bf16 storage (attention arithmetic fp32)",
            "fwd_ms": tf * 1e3, "fwd_bwd_ms": tb * 1e3, "edges_per_s_fwd_bwd": E / tb, "bytes": b_g + b_c, "bound": "hbm", "frac": (b_g + b_c) / tf / HBM_PEAK}


def gat_heads_shard(B=1024, n=16, e=64, F_=200, D=200, H=8, iters=50):
    """cfg4_shard_1gpu: BASELINE.json configs[3]'s per-GPU shard (8 192 graphs / 8 GPUs = 1 024 graphs) as ONE GPU runs it — bench.py's own
    step (H-head attention stage forward + backward, gradients into the flat bucket, no collective).  The base of the scaling curve:
    value(N) of an N-GPU run over N x this figure is shard-for-shard comparable, which value(N) / value(1) of the 512-graph N = 1 line is not."""
    from recon_amd import synth
    from recon_amd.models import SpGAT
    from recon_amd.graph import prepare_graph, clear_graph_cache
    from recon_amd.dist import FlatGradBucket
    dv = torch.device("cuda:0")
    N, E = B * n, B * e
    x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
    torch.manual_seed(0)
    model = SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(dv)
    xd, eed, edged = x.to(dv).requires_grad_(True), ee.to(dv).requires_grad_(True), edge.to(dv)
    nohop = torch.tensor([])
    Gd = torch.randn(N, H * D, generator=torch.Generator().manual_seed(1)).to(dv)
    bucket = FlatGradBucket(model.head_parameters())
    model.write_head_gradients_into(bucket)
    prepare_graph(edged, nohop, N)

    def step():
        bucket.zero()
        xd.grad = None
        eed.grad = None
        model.heads_forward(xd, edged, eed, nohop, nohop).backward(Gd)
        bucket.pack()
    t = _time(step, iters, warm=10)
    clear_graph_cache()
    return {"graphs": B, "N": N, "E": E, "F": F_, "D_per_head": D, "heads": H, "ms_per_step": t * 1e3, "edges_per_s": E / t, "steps": iters,
            "what": "configs[3]'s per-GPU shard (1 024 graphs) on one GPU, bench.py's step without collectives: the shard-for-shard base of a scaling curve"}


def stage_a_iteration(iters=20):
    """N1's caller as the reference runs it (GAT/main.py:478-525): tools/stage_a_iter_bench.py with the fused loss, `iters` fresh iterations."""
    import stage_a_iter_bench
    return stage_a_iter_bench.run(iters=iters, loss_rows="recon", launches=True)


def all_secondary(fast=True):
    it = 6 if fast else 20
    return {"cfg4_shard_1gpu": gat_heads_shard(iters=50),
            "stage_a_iteration": stage_a_iteration(iters=20),
            "cfg3b_n9_propagation": propagation(9, iters=it),
            "cfg3b_n32_propagation": propagation(32, iters=10),          # training legs: max(2, iters // 2) = 5 timed iterations, twice
            "cfg3b_n9_bf16": propagation_bf16(9, iters=it),
            "cfg3b_n32_bf16": propagation_bf16(32, iters=10),
            "cfg3a_gcn_bf16": gcn_bf16(iters=it),
            "cfg5_powerlaw_spgat": powerlaw_spgat(iters=it),
            "cfg5_mixed_stack_bf16": powerlaw_mixed_stack_bf16(iters=it)}


if __name__ == "__main__":
    import json
    torch.autograd.set_multithreading_enabled(False)
    for k, v in all_secondary(fast="--long" not in sys.argv).items():
        print(json.dumps({k: v}))
