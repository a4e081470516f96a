"""Why does `secondary.stage_a_iteration.model_step_distinct_batches_ms` read 5.8 ms inside bench.py and 1.7 ms on its own?  Runs the
stage-A legs after a chosen prefix of what bench.py does before its secondary workloads.

  python tools/probe/stage_a_in_bench_probe.py alone | after_shard | after_main | after_uncached | after_exact
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

import secondary  # noqa: E402

which = sys.argv[1]
dev = torch.device("cuda:0")
if which == "after_shard":
    secondary.gat_heads_shard(iters=10)
if which in ("after_main", "after_uncached", "after_exact"):
    from recon_amd import synth, gat_layers
    from recon_amd.models import SpGAT
    from recon_amd.graph import clear_graph_cache, prepare_graph
    B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
    N = B * n
    x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
    model = SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(dev)
    xd, eed, edged = x.to(dev).requires_grad_(True), ee.to(dev).requires_grad_(True), edge.to(dev)
    nohop = torch.tensor([])
    Gd = torch.randn(N, H * D).to(dev)
    prepare_graph(edged, nohop, N)

    def step(ed=edged):
        xd.grad = None
        eed.grad = None
        model.heads_forward(xd, ed, eed, nohop, nohop).backward(Gd)
    for _ in range(220):
        step()
    torch.cuda.synchronize()
    if which == "after_exact":
        fam = gat_layers._GEMM_BX3
        gat_layers._GEMM_BX3 = "0"
        for _ in range(53):
            step()
        gat_layers._GEMM_BX3 = fam
        torch.cuda.synchronize()
    if which == "after_uncached":
        for ed in [edged.clone() for _ in range(53)]:
            step(ed)
        torch.cuda.synchronize()
        clear_graph_cache()
r = secondary.stage_a_iteration(iters=20)
print(which, {k: round(v, 3) for k, v in r.items() if isinstance(v, float)})
