#!/usr/bin/env python3
"""Launch census of one fresh stage-A iteration (tools/stage_a_iter_bench.py's loop): device kernels / copies / fills per phase
(batch assembly: 1-hop, 2-hop, triples; model forward; loss; backward; optimizer), from torch.profiler.  Runs on the GPU box.
    python3 tools/stage_a_census.py [--iters 5] [--top 12]"""
import argparse, collections, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.stage_a_iter_bench import synthetic_kg, batch_gat_loss
from recon_amd.models import SpKBGATModified
from recon_amd.sampler import KGNeighbourSampler

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("--iters", type=int, default=5); ap.add_argument("--top", type=int, default=12)
    ap.add_argument("--parse", default=None, help="kernel trace CSV of a rocprofv3 --kernel-trace run of this script")
    args = ap.parse_args()
    if args.parse:
        import csv
        rows_ = sorted(csv.DictReader(open(args.parse)), key=lambda r: int(r["Dispatch_Id"]))
        names = ["1-hop batch", "2-hop batch", "triples", "model forward", "loss", "backward", "optimizer + item"]
        size_of = {}
        for i in range(8):
            n = 100003 + 1024 * i
            size_of[n] = i
        def sentinel(r):
            # a fill launch whose element count (grid x workgroup items) matches one of the sentinel buffers
            if "Fill" not in r["Kernel_Name"] and "fill" not in r["Kernel_Name"]:
                return None
            gx = int(r["Grid_Size_X"])
            for n, i in size_of.items():
                if abs(gx * 4 - n) < 1024 or abs(gx - n) < 512 or abs(gx * 16 - n * 4) < 2048:
                    return i
            return None
        cur, counts, iters = None, collections.defaultdict(collections.Counter), 0
        for r in rows_:
            sidx = sentinel(r)
            if sidx is not None:
                cur = sidx if sidx < 7 else None
                iters += sidx == 0
                continue
            if cur is not None:
                counts[cur][r["Kernel_Name"].replace("void ", "").replace("at::native::", "").replace("recon::(anonymous namespace)::", "").replace("(anonymous namespace)::", "")[:80]] += 1
        iters = max(iters - 3, 1)                                       # the three warm-up iterations are in the trace too: average over all
        total_iters = iters + 3
        out, tot = {}, 0.0
        for i in range(7):
            n = sum(counts[i].values()) / total_iters
            tot += n
            out[names[i]] = {"launches_per_iteration": round(n, 1), "top": [(k, round(c / total_iters, 1)) for k, c in counts[i].most_common(args.top)]}
        out["total_per_iteration"] = round(tot, 1)
        print(json.dumps(out, indent=1))
        sys.exit(0)
    dv = torch.device("cuda:0")
    torch.autograd.set_multithreading_enabled(False)
    N, nrel, ratio = 14541, 237, 2
    adj_idx, adj_val = synthetic_kg(N=N, nrel=nrel)
    sampler = KGNeighbourSampler(adj_idx.to(dv), adj_val.to(dv), N)
    torch.manual_seed(0)
    model = SpKBGATModified(torch.randn(N, 50), torch.randn(nrel, 50), [100, 200], [100, 200], 0.3, 0.2, [2, 2]).to(dv)
    model.train()
    opt = torch.optim.SGD(model.parameters(), lr=1e-3)
    loss_fn = torch.nn.MarginRankingLoss(margin=1.0)
    g = torch.Generator().manual_seed(1)
    sources_all = torch.unique(adj_idx[1])
    from recon_amd.graph import trust
    from recon_amd.losses import batch_gat_loss as recon_batch_gat_loss

    # phase boundaries as SENTINEL launches: a fill of a buffer whose size is unique per phase (the kernel trace carries grid sizes); run
    # under `rocprofv3 --kernel-trace` and hand the trace to tools/stage_a_census.py --parse <kernel_trace.csv>
    sentinels = [torch.empty(100003 + 1024 * i, dtype=torch.float32, device=dv) for i in range(8)]

    def mark(i):
        sentinels[i].fill_(1.0)

    def iteration():
        mark(0)
        ents = trust(sources_all[torch.randperm(sources_all.numel(), generator=g)[:128]].to(dv), bound=N)
        (edge, edge_type), (srcs, _) = sampler.batch_adj_data(ents)
        mark(1)
        quads = sampler.batch_nhop_neighbors(srcs)
        mark(2)
        pos = torch.stack((edge[1], edge_type, edge[0]), dim=1)
        neg = pos.repeat(2 * ratio, 1)
        half = neg.shape[0] // 2
        neg[:half, 0] = torch.randint(0, N, (half,), device=dv)
        neg[half:, 2] = torch.randint(0, N, (neg.shape[0] - half,), device=dv)
        train_indices = trust(torch.cat((pos, neg), dim=0), bound=N, rel_bound=nrel)
        mark(3)
        entity_embed, relation_embed, _ = model(None, ents, (edge, edge_type), quads)
        mark(4)
        opt.zero_grad()
        loss = recon_batch_gat_loss(loss_fn, train_indices, entity_embed, relation_embed, valid_invalid_ratio_gat=ratio)
        mark(5)
        loss.backward()
        mark(6)
        opt.step()
        v = loss.data.item()
        mark(7)
        return v
    for _ in range(3):
        iteration()
    torch.cuda.synchronize()
    for _ in range(args.iters):
        iteration()
    torch.cuda.synchronize()
