#!/usr/bin/env python3
"""Where a wave of k_gat_atp_bwd spends its cycles at cfg 2: the library built with EXTRA=-DRECON_K2_STAMPS (tools/ab_builds.sh
"-DRECON_K2_STAMPS" -> librecon_hip_b.so) leaves six s_memtime stamps per wave in its gxd row.  Runs on the GPU box:
    RECON_HIP_LIB=recon_amd/csrc/librecon_hip_b.so python3 tools/probe/k2_stamps.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import torch
from recon_amd import gat_layers, synth
from recon_amd.gat_layers import gat_heads
from recon_amd.graph import prepare_graph

cap = {}
orig = gat_layers._carve


def carve(dev, sizes):
    ws, ptrs = orig(dev, sizes)
    if len(sizes) == 11:
        cap["ws"], cap["ptrs"] = ws, ptrs
    return ws, ptrs


gat_layers._carve = carve
d = torch.device("cuda:0")
workload = sys.argv[1] if len(sys.argv) > 1 else "cfg2"
if workload == "cfg2":
    B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
    N, E = B * n, B * e
    x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
    g = torch.Generator().manual_seed(2)
    a = (torch.randn(H, D, 3 * F_, generator=g) * 0.05).to(d).requires_grad_(True)
    a2 = (torch.randn(H, D, generator=g) * 0.05).to(d).requires_grad_(True)
    G = torch.randn(N, H * D, generator=g).to(d)
    graph = prepare_graph(edge.to(d), None, N)
    xd, eed = x.to(d).requires_grad_(True), ee.to(d).requires_grad_(True)
    for _ in range(5):
        gat_heads(xd, eed, a, a2, graph, None, 0.2, True).backward(G)
else:                                                                    # cfg5f32 / cfg5bf16: the attention layers of tools/secondary.py's power-law legs
    sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
    torch.autograd.set_multithreading_enabled(False)
    import secondary as S
    (S.powerlaw_spgat if workload == "cfg5f32" else S.powerlaw_mixed_stack_bf16)(iters=2)
    F_ = 200
    N = None
torch.cuda.synchronize()
if N is None:
    N = (gat_layers._lib_sizes.__self__ if False else 0) or 0
if not N:                                                                 # size of the gxd slice: up to the next slice of the carve
    N = (int(cap["ptrs"][5]) - int(cap["ptrs"][4])) // (4 * F_)
raw = gat_layers._view_f32(cap["ws"], cap["ptrs"][4], N * F_).view(N, F_)[:, :14].contiguous().cpu().numpy().view(np.uint64)   # [N, 7]
raw = raw[raw[:, 0] != 0]                                                 # (padding rows of the slice; hub nodes' rows are written by their own wave too)
N = raw.shape[0]
st = raw[:, :5].astype(np.int64)
hw = raw[:, 5]
deg = raw[:, 6]
t0 = st[:, 0].min()
dur = st[:, 4] - st[:, 0]
print("nodes", N, "kernel span (cycles of s_memtime)", int(st[:, 4].max() - t0), " mean degree", float(deg.mean()))
names = ["entry -> indices + u staged (barrier)", "-> g_V rows landed, dots with x_i", "-> walk done (stores issued and complete)", "-> end"]
for i in range(4):
    dd = st[:, i + 1] - st[:, i]
    print("  %-46s mean %8.0f  p10 %8.0f  p50 %8.0f  p90 %8.0f" % (names[i], dd.mean(), np.percentile(dd, 10), np.percentile(dd, 50), np.percentile(dd, 90)))
print("  wave total mean %.0f p50 %.0f p90 %.0f" % (dur.mean(), np.percentile(dur, 50), np.percentile(dur, 90)))
start = np.sort(st[:, 0] - t0)
print("  wave start times: p25 %d p50 %d p75 %d p100 %d" % tuple(int(np.percentile(start, q)) for q in (25, 50, 75, 100)))
# waves per SIMD slot: HW_ID bits: wave_id [3:0], simd_id [5:4], cu_id [11:8], sh_id [12], se_id [15:13] (gfx9); xcc in [23:20]?
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; se = (hw >> 13) & 7
key = (hw >> 4) & 0xfffff
u, cnt = np.unique(key, return_counts=True)
print("  distinct (SIMD, CU, SE, ...) keys %d; waves per key: min %d mean %.1f max %d" % (len(u), cnt.min(), cnt.mean(), cnt.max()))
