#!/usr/bin/env python3
"""A/B of one run-time switch of the library on the cfg 2 step, in ONE process on one box (box-to-box spread is +-4 %, larger than most
effects worth measuring): rounds of `steps` steps per value, interleaved, HIP-event timed.
    python3 tools/ab_step.py RECON_ATP_ROW_SCALE 1 0 [--steps 100] [--rounds 5]
    python3 tools/ab_step.py py:_OVERLAP False True          (an attribute of recon_amd.gat_layers)"""
import argparse, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

if __name__ == "__main__":
    ap = argparse.ArgumentParser()
    ap.add_argument("name"); ap.add_argument("values", nargs="+")
    ap.add_argument("--steps", type=int, default=100); ap.add_argument("--rounds", type=int, default=5)
    a = ap.parse_args()
    torch.autograd.set_multithreading_enabled(False)
    from recon_amd import _lib, synth
    from recon_amd.models import SpGAT
    from recon_amd.graph import prepare_graph
    from recon_amd.dist import FlatGradBucket
    dev = torch.device("cuda:0")
    B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
    N = B * n
    x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
    torch.manual_seed(0)
    model = SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(dev)
    xd, eed, edged = x.to(dev).requires_grad_(True), ee.to(dev).requires_grad_(True), edge.to(dev)
    nohop = torch.tensor([])
    Gd = torch.randn(N, H * D, generator=torch.Generator().manual_seed(1)).to(dev)
    bucket = FlatGradBucket(model.head_parameters())
    model.write_head_gradients_into(bucket)
    prepare_graph(edged, nohop, N)

    def step():
        bucket.zero(); xd.grad = None; eed.grad = None
        model.heads_forward(xd, edged, eed, nohop, nohop).backward(Gd)
        bucket.allreduce_mean()
    for _ in range(50):
        step()
    res = {v: [] for v in a.values}
    for r in range(a.rounds):
        for v in a.values:
            if a.name.startswith("py:"):                              # a module attribute of recon_amd.gat_layers (python-side switches)
                from recon_amd import gat_layers
                setattr(gat_layers, a.name[3:], eval(v))
            else:
                _lib.config_set(a.name, None if v == "unset" else v)
            for _ in range(5):
                step()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(a.steps):
                step()
            e1.record()
            torch.cuda.synchronize()
            res[v].append(e0.elapsed_time(e1) / a.steps)
    print(json.dumps({"switch": a.name, "ms_per_step": {v: [round(t, 4) for t in ts] for v, ts in res.items()},
                      "median": {v: sorted(ts)[len(ts) // 2] for v, ts in res.items()}}))
