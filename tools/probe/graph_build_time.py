"""Time of one graph preparation (COO -> CSR + CSC + hub count) from a fresh edge tensor at cfg 2's size; RECON_HIP_LIB selects the build."""
import sys, time
sys.path.insert(0, "/root/repo")
import torch
from recon_amd.graph import prepare_graph, trust
dv = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
B, n, e = 512, 16, 64
N, E = B * n, B * e
base = torch.arange(B).repeat_interleave(e) * n
edges = [torch.stack([torch.randint(0, n, (E,), generator=g) + base, torch.randint(0, n, (E,), generator=g) + base]).to(dv) for _ in range(64)]
for trusted in (False, True):
    for rep in range(3):
        es = [t.clone() for t in edges]
        if trusted:
            for t in es: trust(t, bound=N)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for t in es:
            prepare_graph(t, None, N)
        torch.cuda.synchronize()
        print("trusted" if trusted else "checked", "%.1f us per build" % ((time.perf_counter() - t0) / len(es) * 1e6))
