#!/usr/bin/env python3
"""In-process A/B microbenchmark of the fp32 MFMA GEMM (recon_sgemm): tile configs are switched through
RECON_GEMM_CFG between interleaved rounds, so box-to-box and clock noise cancels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd import _lib

def time_once(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters

def run(M, N, K, nk, cfgs, rounds=5, iters=10):
    d = torch.device("cuda:0")
    A = torch.randn(M, K, device=d)
    B = torch.randn(N, K, device=d) if nk else torch.randn(K, N, device=d)
    Cc = torch.empty(M, N, device=d)
    L = _lib.lib(); st = _lib.current_stream()
    def mine():
        L.recon_sgemm(M, N, K, A.data_ptr(), K, B.data_ptr(), B.shape[1], nk, Cc.data_ptr(), N, st)
    def ref():
        torch.mm(A, B.t() if nk else B)
    ws = torch.empty(L.recon_sgemm_bx3_workspace_bytes(N, K), dtype=torch.uint8, device=d) if nk else None
    def bx3():
        L.recon_sgemm_bx3(M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N, ws.data_ptr(), st)
    res = {c: [] for c in cfgs}; res["torch"] = []
    if nk: res["bx3"] = []
    for r in range(rounds + 1):
        for c in cfgs:
            _lib.config_set("RECON_GEMM_CFG", str(c))
            t = time_once(mine, iters)
            if r: res[c].append(t)
        t = time_once(ref, iters)
        if r: res["torch"].append(t)
        if nk:
            t = time_once(bx3, iters)
            if r: res["bx3"].append(t)
    fl = 2.0 * M * N * K
    line = "M=%6d N=%5d K=%6d nk=%d :" % (M, N, K, nk)
    for c in list(cfgs) + ["torch"] + (["bx3"] if nk else []):
        med = sorted(res[c])[len(res[c]) // 2]
        line += "  [%s] %7.1f us %6.1f TF" % (c, med, fl / med / 1e6)
    print(line)

if __name__ == "__main__":
    cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else ["1", "3", "4"])]
    for shp in [(4096, 4096, 4096, 1), (8192, 3200, 200, 1), (65536, 200, 600, 1), (65536, 200, 600, 0), (8192, 600, 200, 0),
                (32768, 200, 1600, 0), (32768, 208, 1600, 0), (65536, 600, 200, 1), (65536, 624, 224, 1)]:
        run(*shp, cfgs=cfgs)
