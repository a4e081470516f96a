#!/usr/bin/env python3
"""One GEMM shape, few launches: run under rocprofv3 --pmc to read SQ counters of the GEMM kernels."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd import _lib
M, N, K, nk = [int(v) for v in (sys.argv[1:5] if len(sys.argv) >= 5 else (4096, 4096, 4096, 1))]
d = torch.device("cuda:0")
A = torch.randn(M, K, device=d)
B = torch.randn(N, K, device=d) if nk else torch.randn(K, N, device=d)
Cc = torch.empty(M, N, device=d)
L = _lib.lib(); st = _lib.current_stream()
for _ in range(5):
    L.recon_sgemm(M, N, K, A.data_ptr(), K, B.data_ptr(), B.shape[1], nk, Cc.data_ptr(), N, st)
torch.cuda.synchronize()
