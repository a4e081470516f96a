"""Does the alignment of V's rows cost K1' write bandwidth?  The training forward of the 8-head attention stage at cfg 2's node / edge counts
with F = R in {192, 200, 208, 224, 256}: a half-term row of V is 2 (2F + R) bytes = 1152 (9 x 128), 1200 (9.375 x 128), 1248, 1344, 1536 bytes.
Run under rocprofv3 --kernel-trace --stats (tools/probe/k1_row_alignment.sh) and read k_gat_atp_fwd's time per width; bytes written per launch
= N (H (F + R) + F) 4."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from recon_amd import synth as O
from recon_amd import gat_layers
from recon_amd.graph import prepare_graph

B, n, e, D, H = 512, 16, 64, 200, 8
d = torch.device("cuda:0")
gat_layers._GAT_PATH = "atp"
for F_ in (192, 200, 208, 224, 256):
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
    g = torch.Generator().manual_seed(0)
    a = torch.stack([O.xavier_normal((D, 3 * F_), 1.414, g) for _ in range(H)]).to(d).requires_grad_(True)
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)]).to(d).requires_grad_(True)
    xd, eed = x.to(d).requires_grad_(True), ee.to(d).requires_grad_(True)
    graph = prepare_graph(edge.to(d), None, B * n)
    for _ in range(3):
        o = gat_layers.gat_heads(xd, eed, a, a2, graph, None, 0.2, True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20 + F_ - 192):                                      # the launch count identifies the width in the kernel trace
        o = gat_layers.gat_heads(xd, eed, a, a2, graph, None, 0.2, True)
    e1.record(); torch.cuda.synchronize()
    N = B * n
    print("F = R = %d: forward %.1f us per call; V bytes written %.1f MB" % (F_, e0.elapsed_time(e1) * 1e3 / (20 + F_ - 192), N * (H * 2 * F_ + F_) * 4 / 1e6))
