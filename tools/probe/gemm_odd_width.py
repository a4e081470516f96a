import sys, torch
sys.path.insert(0, "/root/repo")
from recon_amd import _lib
d = torch.device("cuda:0"); L = _lib.lib(); st = _lib.current_stream()
def t(fn, it=50):
    for _ in range(5): fn()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1) * 1e3 / it
for (M, N, K, nk) in ((14541, 100, 150, 1), (14541, 100, 152, 1), (14541, 150, 100, 1), (14541, 152, 100, 1), (14541, 100, 150, 0), (14541, 200, 150, 0), (29082, 100, 150, 1), (29082, 100, 152, 1)):
    A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d) if nk else torch.randn(K, N, device=d); C = torch.empty(M, N, device=d)
    f = lambda: L.recon_sgemm(M, N, K, A.data_ptr(), K, B.data_ptr(), B.shape[1], nk, C.data_ptr(), N, st)
    g = lambda: torch.mm(A, B.t() if nk else B, out=C)
    print(M, N, K, "nk" if nk else "kn", "recon %.1f us   torch.mm %.1f us" % (t(f), t(g)))
