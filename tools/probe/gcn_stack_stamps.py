"""Cycle stamps of one wave of the fused GraphConvolution stack (library built with EXTRA=-DRECON_STAMPS): wait / barrier times per K step."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
from recon_amd.gcn_layers import GraphConvolution, gcn_stack
dv = torch.device("cuda:0")
n, D, B, L = 32, 300, 1024, 3
g = torch.Generator().manual_seed(0)
x = torch.randn(B, n, D, generator=g).to(torch.bfloat16).to(dv)
adj = (torch.rand(B, n, n, generator=g) < 0.2).to(torch.bfloat16).to(dv)
torch.manual_seed(0)
layers = [GraphConvolution(D, D).to(dv).to(torch.bfloat16) for _ in range(L)]
with torch.no_grad():
    for _ in range(3):
        out = gcn_stack(x, adj, layers)
    torch.cuda.synchronize()
raw = out[0, 0, :192].contiguous().view(torch.int16).cpu().numpy().view("uint64")
ns = int(raw[47])
st = raw[:ns].astype("int64")
print("stamps", ns, "total cycles", st[-1] - st[0])
d = st[1:] - st[:-1]
print("prologue->first wait", d[0])
pairs = d[1:-1]
print("per step: [barrier wait, copy issue, fragment reads landed, matrix pipe + wait for this wave's copies]")
for i in range(0, len(pairs) - 3, 4):
    print([int(v) for v in pairs[i:i + 4]])
