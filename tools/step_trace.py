import os, sys, time, torch
sys.path.insert(0, os.getcwd())
from recon_amd.models import SpGAT
from recon_amd.graph import prepare_graph
from recon_amd.dist import FlatGradBucket
from recon_amd import synth
dev = torch.device("cuda:0")
B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
N, E = B*n, B*e
x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
torch.manual_seed(0)
model = SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(dev)
xd = x.to(dev).requires_grad_(True); eed = ee.to(dev).requires_grad_(True); edged = edge.to(dev)
nohop = torch.tensor([]); Gd = torch.randn(N, H*D, generator=torch.Generator().manual_seed(1)).to(dev)
bucket = FlatGradBucket([p for att in model.attentions for p in (att.a, att.a_2)])
graph = prepare_graph(edged, nohop, N)
def step():
    bucket.zero(); xd.grad = None; eed.grad = None
    out = model.heads_forward(xd, edged, eed, nohop, nohop); out.backward(Gd); bucket.allreduce_mean()
ts = []
for i in range(80):
    torch.cuda.synchronize(); t0 = time.perf_counter(); step(); torch.cuda.synchronize(); ts.append((time.perf_counter()-t0)*1e3)
print("per-step sync'd ms:", " ".join("%.3f" % t for t in ts[:12]), "...", " ".join("%.3f" % t for t in ts[-6:]))
# unsynchronised: CPU launch time of a step
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): step()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("50 steps: cpu enqueue %.3f ms/step, total %.3f ms/step" % ((t1-t0)*20, (t2-t0)*20))
# ---- where the CPU time goes: the two C calls alone (buffers allocated once), enqueue only
import ctypes as C
from recon_amd import _lib
from recon_amd.gat_layers import _atp_args, _atp_split_buffer
L = _lib.lib(); st = _lib.current_stream(); f32 = dict(dtype=torch.float32, device=dev)
with torch.no_grad():
    a = torch.stack([att.a for att in model.attentions]).contiguous(); a2 = torch.cat([att.a_2 for att in model.attentions], 0).contiguous()
W = 3 * F_
out = torch.empty(N, H*D, **f32); u = torch.empty(H, W, **f32); c_node = torch.empty(N, 2*H, **f32); c_rel = torch.empty(E, H, **f32)
V = torch.empty(N, H, W, **f32); sigma = torch.empty(E, H, **f32); Z = torch.empty(N, H, **f32); Zk = torch.empty(N, H, **f32)
a_split, aux = _atp_split_buffer(F_, F_, D, H, dev, N)
fa = _atp_args(graph, xd.detach(), eed.detach(), a, a2, None, u, c_node, c_rel, V, sigma, Z, Zk, out, 0.2, True, a_split, aux)
g_V = torch.empty(N, H, W, **f32); g_sigma = torch.empty(E, H, **f32); Gxs = torch.empty(E, F_, **f32); gxd = torch.empty(N, F_, **f32)
Gs = torch.empty(N, 2*H, **f32); g_u = torch.empty(H, W, **f32); q = torch.empty(N, H, **f32)
partial = torch.empty(L.recon_gat_atp_bwd_partial_floats(N, E, F_, F_, D, H), **f32); partial2 = torch.empty(L.recon_gat_atp_bwd_partial2_floats(N, E, F_, F_, D, H), **f32)
g_x = torch.empty(N, F_, **f32); g_ee = torch.empty(E, F_, **f32); g_a = torch.empty(H, D, W, **f32); g_a2 = torch.empty(H, D, **f32)
gh_split = torch.empty(L.recon_gat_atp_bwd_split_bytes(N, D, H), dtype=torch.uint8, device=dev)
ba = _lib.GatAtpBwdArgs(fa, Gd.data_ptr(), H*D, None, g_V.data_ptr(), g_sigma.data_ptr(), Gxs.data_ptr(), gxd.data_ptr(), Gs.data_ptr(), g_u.data_ptr(),
                        q.data_ptr(), partial.data_ptr(), partial2.data_ptr(), g_x.data_ptr(), g_ee.data_ptr(), g_a.data_ptr(), g_a2.data_ptr(), gh_split.data_ptr())
def cstep():
    L.recon_gat_atp_fwd(C.byref(graph.c), C.byref(fa), st); L.recon_gat_atp_bwd(C.byref(graph.c), C.byref(ba), st)
for _ in range(5): cstep()
torch.cuda.synchronize(); t0 = time.perf_counter()
for i in range(50): cstep()
t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
print("C calls only: cpu enqueue %.3f ms/step, total %.3f ms/step" % ((t1-t0)*20, (t2-t0)*20))
