#!/usr/bin/env python3
"""Microbenchmark of the split-precision GEMM families on the layer's three product shapes (cfg 2):
bf16 x 3 (A split on the fly) vs f16 x 2 (both operands pre-split; GEMM alone) vs torch.mm fp32."""
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


def aligned(nbytes, d):
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=d)
    return ws, ws.data_ptr() + ((-ws.data_ptr()) % 256)


def run_nk(M, N, K, rounds=5, iters=10):
    d = torch.device("cuda:0")
    A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d) * 0.07
    Cc = torch.empty(M, N, device=d)
    L = _lib.lib(); st = _lib.current_stream()
    ws3 = torch.empty(L.recon_sgemm_bx3_workspace_bytes(N, K), dtype=torch.uint8, device=d)
    keep, ws2 = aligned(L.recon_sgemm_hx2_workspace_bytes(M, N, K), d)
    L.recon_sgemm_hx2(M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N, ws2, st)
    ref = A.double() @ B.double().t()
    e_h = (Cc.double() - ref).abs().max().item()
    fns = {"torch": lambda: torch.mm(A, B.t()),
           "bx3": lambda: L.recon_sgemm_bx3(M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N, ws3.data_ptr(), st),
           "hx2": lambda: L.recon_sgemm_hx2_presplit(M, N, K, Cc.data_ptr(), N, ws2, st),
           "hx2+split": lambda: L.recon_sgemm_hx2(M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N, ws2, st)}
    report("nk", M, N, K, fns, rounds, iters, e_h)


def run_tn(M, N, K, rounds=5, iters=10):
    d = torch.device("cuda:0")
    A = torch.randn(K, M, device=d); B = torch.randn(K, N, device=d)
    Cc = torch.empty(M, N, device=d)
    L = _lib.lib(); st = _lib.current_stream()
    ws3 = torch.empty(L.recon_sgemm_bx3_tn_workspace_bytes(M, N, K), dtype=torch.uint8, device=d)
    keep, ws2 = aligned(L.recon_sgemm_hx2_tn_workspace_bytes(M, N, K), d)
    L.recon_sgemm_hx2_tn(M, N, K, A.data_ptr(), M, B.data_ptr(), N, Cc.data_ptr(), N, ws2, st)
    ref = A.double().t() @ B.double()
    e_h = (Cc.double() - ref).abs().max().item()
    fns = {"torch": lambda: torch.mm(A.t(), B),
           "bx3": lambda: L.recon_sgemm_bx3_tn(M, N, K, A.data_ptr(), M, B.data_ptr(), N, Cc.data_ptr(), N, ws3.data_ptr(), st),
           "hx2": lambda: L.recon_sgemm_hx2_tn_presplit(M, N, K, Cc.data_ptr(), N, ws2, st)}
    report("tn", M, N, K, fns, rounds, iters, e_h)


def report(kind, M, N, K, fns, rounds, iters, err):
    res = {k: [] for k in fns}
    for r in range(rounds + 1):
        for k, f in fns.items():
            t = time_once(f, iters)
            if r:
                res[k].append(t)
    fl = 2.0 * M * N * K
    line = "%s M=%6d N=%5d K=%6d :" % (kind, M, N, K)
    for k in fns:
        med = sorted(res[k])[len(res[k]) // 2]
        line += "  [%s] %7.1f us %6.1f TF" % (k, med, fl / med / 1e6)
    print(line + "   hx2 max|err| %.2e" % err, flush=True)


if __name__ == "__main__":
    for shp in [(65536, 200, 600), (65536, 600, 200), (4096, 4096, 4096), (8192, 1600, 4800)]:
        run_nk(*shp)
    for shp in [(600, 1600, 8192), (4800, 1600, 8192), (4096, 4096, 4096)]:
        run_tn(*shp)
