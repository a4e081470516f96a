#!/usr/bin/env python3
"""Microbenchmark of the fp32 MFMA GEMM (recon_sgemm) on the shapes of the GAT projections."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd import _lib

def run(M, N, K, nk, iters=20):
    d = torch.device("cuda:0")
    A = torch.randn(M, K, device=d)
    B = torch.randn(N, K, device=d) if nk else torch.randn(K, N, device=d)
    Cc = torch.empty(M, N, device=d)
    L = _lib.lib(); st = _lib.current_stream()
    for _ in range(3):
        L.recon_sgemm(M, N, K, A.data_ptr(), K, B.data_ptr(), B.shape[1], nk, Cc.data_ptr(), N, st)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        L.recon_sgemm(M, N, K, A.data_ptr(), K, B.data_ptr(), B.shape[1], nk, Cc.data_ptr(), N, st)
    e1.record(); torch.cuda.synchronize()
    us = e0.elapsed_time(e1) * 1e3 / iters
    ref = A[:64] @ (B.t() if nk else B)
    err = (Cc[:64] - ref).abs().max().item()
    t0 = time.perf_counter()
    for _ in range(iters):
        R = A @ (B.t() if nk else B)
    torch.cuda.synchronize()
    us_t = (time.perf_counter() - t0) * 1e6 / iters
    print("M=%6d N=%5d K=%6d nk=%d : %8.1f us  %6.1f TF/s   (torch.mm %8.1f us %6.1f TF/s)  err %.1e" % (
        M, N, K, nk, us, 2.0 * M * N * K / us / 1e6, us_t, 2.0 * M * N * K / us_t / 1e6, err))

if __name__ == "__main__":
    print("cfg", os.environ.get("RECON_GEMM_CFG"), "splitk", os.environ.get("RECON_GEMM_SPLITK"))
    for shp in [(4096, 4096, 4096, 1), (8192, 3200, 200, 1), (32768, 1600, 200, 1), (8192, 200, 3200, 0),
                (32768, 200, 1600, 0), (32768, 224, 1600, 0), (32768, 256, 1600, 0)]:
        run(*shp)
