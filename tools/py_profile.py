"""cProfile of the Python side of one bench step (where the 0.45 ms of CPU time per step goes)."""
import cProfile, pstats, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd.models import SpGAT
from recon_amd.graph import prepare_graph
from recon_amd.dist import FlatGradBucket
from recon_amd import synth
dev = torch.device("cuda:0")
B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
N, E = B * n, B * e
x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
torch.manual_seed(0)
model = SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(dev)
xd = x.to(dev).requires_grad_(True); eed = ee.to(dev).requires_grad_(True); edged = edge.to(dev)
nohop = torch.tensor([]); Gd = torch.randn(N, H * D, generator=torch.Generator().manual_seed(1)).to(dev)
bucket = FlatGradBucket([p for att in model.attentions for p in (att.a, att.a_2)])
graph = prepare_graph(edged, nohop, N)
def step():
    bucket.zero(); xd.grad = None; eed.grad = None
    out = model.heads_forward(xd, edged, eed, nohop, nohop); out.backward(Gd); bucket.allreduce_mean()
for _ in range(20): step()
torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
for _ in range(200): step()
pr.disable(); torch.cuda.synchronize()
pstats.Stats(pr).sort_stats("tottime").print_stats(22)
# ---- split of the backward: our Function.backward body vs the rest of the autograd engine call
import time
from recon_amd import gat_layers
orig = gat_layers._GATHeadsATPFunction.backward
acc = {"body": 0.0, "n": 0}
def timed(ctx, g):
    t0 = time.perf_counter(); r = orig(ctx, g); acc["body"] += time.perf_counter() - t0; acc["n"] += 1; return r
gat_layers._GATHeadsATPFunction.backward = staticmethod(timed)
tf = tb = tz = 0.0
for _ in range(200):
    t0 = time.perf_counter(); bucket.zero(); xd.grad = None; eed.grad = None
    t1 = time.perf_counter(); out = model.heads_forward(xd, edged, eed, nohop, nohop)
    t2 = time.perf_counter(); out.backward(Gd)
    t3 = time.perf_counter(); tz += t1 - t0; tf += t2 - t1; tb += t3 - t2
torch.cuda.synchronize()
print("per step (us): zero %.1f  forward %.1f  backward call %.1f  of which Function.backward body %.1f (n=%d)" % (tz * 5e3, tf * 5e3, tb * 5e3, acc["body"] * 5e3, acc["n"]))
# ---- cProfile of the Function bodies alone, called on the main thread
gat_layers._GATHeadsATPFunction.backward = staticmethod(orig)
from recon_amd.gat_layers import gat_heads
a_f, a2_f = model.fused_head_params()
out = gat_heads(xd, eed, a_f, a2_f, graph, None, 0.2, True)
fn = out.grad_fn
for _ in range(10): gat_layers._GATHeadsATPFunction.backward(fn, Gd)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): gat_layers._GATHeadsATPFunction.backward(fn, Gd)
pr.disable(); torch.cuda.synchronize()
print("==== backward body"); pstats.Stats(pr).sort_stats("tottime").print_stats(14)
pr = cProfile.Profile(); pr.enable()
for _ in range(200): model.heads_forward(xd, edged, eed, nohop, nohop)
pr.disable(); torch.cuda.synchronize()
print("==== heads_forward"); pstats.Stats(pr).sort_stats("tottime").print_stats(16)
