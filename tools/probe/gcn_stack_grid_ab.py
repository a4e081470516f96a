"""gcn_stack forward at cfg 3a with the wave grids of k_gcn_b16_stack_fwd: RECON_GCN_STACK_GPW = 1 (4 column parts x 4 graph slots, 16 waves),
2 (4 x 2, 8 waves), 5 (5 x 2, 10 waves).  Bit-equal outputs; times from HIP events."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recon_amd import _lib
from recon_amd.gcn_layers import GraphConvolution, gcn_stack
dv = torch.device("cuda:0")
n, D, B, L = 32, 300, 1024, 3
g = torch.Generator().manual_seed(0)
x = torch.randn(B, n, D, generator=g).to(torch.bfloat16).to(dv)
adj = (torch.rand(B, n, n, generator=g) < 0.2).to(torch.bfloat16).to(dv)
torch.manual_seed(0)
layers = [GraphConvolution(D, D).to(dv).to(torch.bfloat16) for _ in range(L)]
ref = None
for rnd in range(2):
    for gpw in ("1", "2", "5"):
        _lib.config_set("RECON_GCN_STACK_GPW", gpw)
        with torch.no_grad():
            for _ in range(5):
                out = gcn_stack(x, adj, layers)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(50):
                out = gcn_stack(x, adj, layers)
            e1.record(); torch.cuda.synchronize()
        if ref is None:
            ref = out.clone()
        print("GPW %s: %.1f us  bit-equal %s" % (gpw, e0.elapsed_time(e1) * 1e3 / 50, bool(torch.equal(out, ref))))
