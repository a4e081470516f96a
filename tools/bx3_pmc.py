#!/usr/bin/env python3
"""Few launches of the split-precision GEMMs (cfg-2 shapes) for rocprofv3 --pmc runs."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from recon_amd import _lib
d = torch.device("cuda:0")
L = _lib.lib(); st = _lib.current_stream()
M, N, K = 65536, 200, 600
A = torch.randn(M, K, device=d); B = torch.randn(N, K, device=d); Cc = torch.empty(M, N, device=d)
ws = torch.empty(L.recon_sgemm_bx3_workspace_bytes(N, K), dtype=torch.uint8, device=d)
for _ in range(5):
    L.recon_sgemm_bx3(M, N, K, A.data_ptr(), K, B.data_ptr(), K, Cc.data_ptr(), N, ws.data_ptr(), st)
M2, N2, K2 = 600, 200, 65536
A2 = torch.randn(K2, M2, device=d); B2 = torch.randn(K2, N2, device=d); C2 = torch.empty(M2, N2, device=d)
ws2 = torch.empty(L.recon_sgemm_bx3_tn_workspace_bytes(M2, N2, K2), dtype=torch.uint8, device=d)
for _ in range(5):
    L.recon_sgemm_bx3_tn(M2, N2, K2, A2.data_ptr(), M2, B2.data_ptr(), N2, C2.data_ptr(), N2, ws2.data_ptr(), st)
torch.cuda.synchronize()
