#!/usr/bin/env python3
"""Launch-bound regimes (SURVEY 7 "hard parts": hipGraph or equivalent): the cfg 1 step — the reference's own CPU-runnable case, B = 32 graphs of 8
nodes / 56 edges, F = R = D = 50, 1 head: 14 launches for ~30 us of device work — captured ONCE into a HIP graph (torch.cuda.CUDAGraph: every op of
this package launches on torch's current stream, allocates through torch's allocator and — with a cached graph structure — never reads back to
the host, so forward + backward of the attention layer is capturable as it is) and replayed.  Prints eager vs replay time per step.

  python tools/graph_replay_bench.py [--cfg 1|2] [--iters 200]
"""
import argparse, json, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd import synth
from recon_amd.gat_layers import SpGraphAttentionLayer
from recon_amd.models import SpGAT
from recon_amd.graph import prepare_graph


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cfg", type=int, default=1)
    ap.add_argument("--iters", type=int, default=200)
    a = ap.parse_args()
    torch.autograd.set_multithreading_enabled(False)
    dv = torch.device("cuda:0")
    if a.cfg == 1:
        B, n, e, F_, D, H = 32, 8, 56, 50, 50, 1
    else:
        B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
    N = B * n
    x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
    torch.manual_seed(0)
    model = SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(dv)
    xd, eed, ed = x.to(dv).requires_grad_(True), ee.to(dv).requires_grad_(True), edge.to(dv)
    nohop = torch.tensor([])
    G = torch.randn(N, H * D, generator=torch.Generator().manual_seed(1)).to(dv)
    params = [p for att in model.attentions for p in (att.a, att.a_2)]
    prepare_graph(ed, nohop, N)

    def step():
        for p in params:
            p.grad = None
        xd.grad = None
        eed.grad = None
        model.heads_forward(xd, ed, eed, nohop, nohop).backward(G)

    def timed(fn, iters):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(iters):
            fn()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / iters * 1e3

    for _ in range(5):
        step()
    t_eager = timed(step, a.iters)
    ref = [p.grad.clone() for p in params] + [xd.grad.clone()]
    # capture: a side stream (capture needs a non-default stream), three warm-up steps on it, then one captured step
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        step()
    g.replay()
    torch.cuda.synchronize()
    same = all(torch.equal(p.grad, r) for p, r in zip(params, ref[:-1])) and torch.equal(xd.grad, ref[-1])
    t_replay = timed(g.replay, a.iters)
    print(json.dumps({"workload": "cfg %d attention stage fwd+bwd (heads only)" % a.cfg, "eager_ms_per_step": t_eager, "graph_replay_ms_per_step": t_replay,
                      "speedup": t_eager / t_replay, "replay_gradients_bit_equal_to_eager": bool(same)}))


if __name__ == "__main__":
    main()
