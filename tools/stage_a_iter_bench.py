#!/usr/bin/env python3
"""One stage-A (KB-GAT) training iteration as the reference runs it — /root/reference/GAT/main.py:478-525: per iteration a FRESH
entity batch, its 1-hop adjacency (`Corpus.get_batch_adj_data`) and 2-hop quadruples (`get_batch_nhop_neighbors_all`) as NEW tensors,
`SpKBGATModified.forward`, the TransE margin loss of `batch_gat_loss` (:344-376, valid_invalid_ratio_gat = 2), backward, SGD step and
the `loss.item()` read-back — on an FB15k-237-sized synthetic knowledge graph (14 541 entities, 237 relations, 272 115 triples, Zipf-like
degrees), 128 entities per batch (`--entities_per_batch` of the reference's run scripts).

Nothing keyed on tensor identity survives an iteration in this regime ("fresh").  For comparison the same iteration with ONE batch
re-used every time ("cached": what tools/kg_scale_bench.py --kbgat measures) and the batch assembly alone.  Prints one JSON line.

  python tools/stage_a_iter_bench.py [--iters 30] [--entities 128] [--no-2hop]
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.models import SpKBGATModified  # noqa: E402
from recon_amd.sampler import KGNeighbourSampler  # noqa: E402


def synthetic_kg(N=14541, T=272115, nrel=237, seed=0):
    rs = np.random.RandomState(seed)
    p = 1.0 / np.arange(1, N + 1) ** 0.8
    p /= p.sum()
    tails = rs.permutation(N)[rs.choice(N, size=T, p=p)]
    heads = rs.permutation(N)[rs.choice(N, size=T, p=p)]
    rel = rs.randint(0, nrel, size=T)
    return torch.from_numpy(np.stack([tails, heads])).long(), torch.from_numpy(rel).long()


def batch_gat_loss(loss_fn, train_indices, entity_embed, relation_embed, ratio=2, rows=None):
    """GAT/main.py:344-376 restated: positives first, 2 * ratio negatives per positive behind them.  `rows(table, index)`: how the
    embedding rows are fetched — None = `table[index]` as the reference writes it (its backward is torch's sort-based
    indexing_backward_kernel: 0.33 ms per gather at these sizes, six gathers per iteration), or recon_amd's gather_rows (fixed-order
    segment sums, the SpecialSpmmFinal walk)."""
    if rows is None:
        rows = lambda table, index: table[index]
    n_pos = train_indices.shape[0] // (2 * ratio + 1)
    pos = train_indices[:n_pos].repeat(2 * ratio, 1)
    neg = train_indices[n_pos:]
    pos_norm = torch.norm(rows(entity_embed, pos[:, 0]) + rows(relation_embed, pos[:, 1]) - rows(entity_embed, pos[:, 2]), p=1, dim=1)
    neg_norm = torch.norm(rows(entity_embed, neg[:, 0]) + rows(relation_embed, neg[:, 1]) - rows(entity_embed, neg[:, 2]), p=1, dim=1)
    y = -torch.ones(2 * ratio * n_pos, device=entity_embed.device)
    return loss_fn(pos_norm, neg_norm, y)


def count_launches(fn, iters=3):
    """Device launches (kernels, copies, fills) per call of `fn`, from torch.profiler's device-side events; None where the profiler
    records none (no roctracer in the build)."""
    try:
        from torch.profiler import profile, ProfilerActivity
        torch.cuda.synchronize()
        with profile(activities=[ProfilerActivity.CUDA]) as prof:
            for _ in range(iters):
                fn()
            torch.cuda.synchronize()
        n = sum(1 for ev in prof.events() if str(getattr(ev, "device_type", "")).endswith("CUDA"))
        return round(n / iters, 1) if n else None
    except Exception:
        return None


def main(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--entities", type=int, default=128)
    ap.add_argument("--no-2hop", action="store_true")
    ap.add_argument("--ratio", type=int, default=2)
    ap.add_argument("--loss-rows", choices=["torch", "recon"], default="torch", help="table[index] (the reference's loss code) or recon_amd.gat_layers.gather_rows")
    ap.add_argument("--profile", action="store_true", help="cProfile of the host side of fresh iterations (top functions by own time)")
    ap.add_argument("--launches", action="store_true", help="also count the device launches of a fresh iteration (torch.profiler)")
    args = ap.parse_args(argv)
    dv = torch.device("cuda:0")
    torch.autograd.set_multithreading_enabled(False)
    N, nrel = 14541, 237
    adj_idx, adj_val = synthetic_kg(N=N, nrel=nrel)
    sampler = KGNeighbourSampler(adj_idx.to(dv), adj_val.to(dv), N)
    torch.manual_seed(0)
    model = SpKBGATModified(torch.randn(N, 50), torch.randn(nrel, 50), [100, 200], [100, 200], 0.3, 0.2, [2, 2]).to(dv)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    loss_fn = torch.nn.MarginRankingLoss(margin=1.0)
    g = torch.Generator().manual_seed(1)
    sources_all = torch.unique(adj_idx[1])

    rows = None
    fused_loss = args.loss_rows == "recon"                           # recon_amd.losses.batch_gat_loss: the reference's function as one fused op
    from recon_amd.graph import trust
    from recon_amd.losses import batch_gat_loss as recon_batch_gat_loss

    def make_batch():
        ents = trust(sources_all[torch.randperm(sources_all.numel(), generator=g)[:args.entities]].to(dv), bound=N)     # drawn from the graph's own sources
        (edge, edge_type), (srcs, _) = sampler.batch_adj_data(ents)
        quads = torch.tensor([], dtype=torch.long) if args.no_2hop else sampler.batch_nhop_neighbors(srcs)
        # training triples of the batch (head, relation, tail) + 2 * ratio corrupted copies (Corpus.get_iteration_triples_batch)
        pos = torch.stack((edge[1], edge_type, edge[0]), dim=1)
        neg = pos.repeat(2 * args.ratio, 1)
        half = neg.shape[0] // 2
        neg[:half, 0] = torch.randint(0, N, (half,), device=dv)
        neg[half:, 2] = torch.randint(0, N, (neg.shape[0] - half,), device=dv)
        # ids of the sampler's edges + randint(0, N): in range by construction (the loss does not validate them with a host round trip)
        return ents, (edge, edge_type), quads, trust(torch.cat((pos, neg), dim=0), bound=N, rel_bound=nrel)

    def train_iter(batch):
        ents, adj, quads, train_indices = batch
        entity_embed, relation_embed, _ = model(None, ents, adj, quads)
        opt.zero_grad()
        if fused_loss:
            loss = recon_batch_gat_loss(loss_fn, train_indices, entity_embed, relation_embed, valid_invalid_ratio_gat=args.ratio)
        else:
            loss = batch_gat_loss(loss_fn, train_indices, entity_embed, relation_embed, args.ratio, rows)
        loss.backward()
        opt.step()
        return loss.data.item()                                       # the reference reads the loss back every iteration

    def timed(fn, iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    one = make_batch()
    for _ in range(3):
        train_iter(one)
        train_iter(make_batch())
    if args.profile:
        import cProfile, pstats
        pr = cProfile.Profile(); pr.enable()
        for _ in range(args.iters):
            train_iter(make_batch())
        pr.disable(); torch.cuda.synchronize()
        st = pstats.Stats(pr); st.sort_stats("tottime").print_stats(30); st.sort_stats("cumtime").print_stats(70)
        return
    # The cyclic garbage collector is held off over the timed legs: inside bench.py (a process that has built and dropped the main workload's
    # tensors, autograd graphs and ctypes structs) a generation-2 collection landed in one of these 20-iteration legs in every run and cost it
    # ~80 ms — 4 ms per iteration of whichever leg it hit (round 6: the distinct-batch leg read 5.8 ms in BENCH, 1.7 on its own)
    import gc
    gc.collect()
    gc.disable()
    t_batch = timed(make_batch, args.iters)
    each_cached = []

    def one_cached():
        t0 = time.perf_counter()
        train_iter(one)
        each_cached.append((time.perf_counter() - t0) * 1e3)
    t_cached = timed(one_cached, args.iters)
    each_fresh = []

    def one_fresh():                                                  # ends in loss.item(): every iteration is its own timed unit
        t0 = time.perf_counter()
        train_iter(make_batch())
        each_fresh.append((time.perf_counter() - t0) * 1e3)
    t_fresh = timed(one_fresh, args.iters)
    # the model step alone on batches it has not seen: the batches are assembled BEFORE the clock starts, so the
    # figure is independent of how much of the assembly hides behind the previous step's device work
    distinct = [make_batch() for _ in range(args.iters)]
    it = iter(distinct)
    each = []

    def one_distinct():                                               # train_iter ends in loss.item(): every iteration is its own timed unit
        t0 = time.perf_counter()
        train_iter(next(it))
        each.append((time.perf_counter() - t0) * 1e3)
    t_distinct = timed(one_distinct, args.iters)
    gc.enable()
    E1, E2 = one[1][0].shape[1], (0 if args.no_2hop else one[2].shape[0])
    launches = count_launches(lambda: train_iter(make_batch())) if args.launches else None
    res = ({"workload": "stage-A iteration (GAT/main.py:478-525): sampler batch -> SpKBGATModified fwd -> margin loss -> bwd -> SGD, FB15k-237-sized synthetic KG",
                      "entities_per_batch": args.entities, "loss_rows": args.loss_rows, "edges_1hop": E1, "quads_2hop": E2,
                      "batch_assembly_ms": t_batch, "iteration_cached_batch_ms": t_cached, "iteration_cached_median_ms": float(np.median(each_cached)),
                      "iteration_fresh_batch_ms": t_fresh,
                      "iteration_fresh_median_ms": float(np.median(each_fresh)), "iteration_fresh_max_ms": float(max(each_fresh)),
                      "model_step_distinct_batches_ms": t_distinct, "model_step_distinct_median_ms": float(np.median(each)),
                      "model_step_distinct_max_ms": float(max(each)), "distinct_over_cached": t_distinct / t_cached,
                      "assembly_hidden_ms": t_batch + t_distinct - t_fresh,
                      "edges_per_s_fresh": (E1 + E2) / t_fresh * 1e3, "iters": args.iters,
                      "launches_per_fresh_iteration": launches})
    if argv is None:
        print(json.dumps(res))
    return res


def run(iters=20, loss_rows="recon", launches=True):
    """The bench line's `secondary.stage_a_iteration` (bench.py): `iters` fresh iterations per leg."""
    return main(["--iters", str(iters), "--loss-rows", loss_rows] + (["--launches"] if launches else []))


if __name__ == "__main__":
    main()
