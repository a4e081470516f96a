"""Ablations of the batched bf16 GEMM on the n = 32 forward (timing only; results are garbage under RECON_BGEMM_ABL)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.secondary import _time
from recon_amd.propagation import propagate, get_head_indices, get_tail_indices

if __name__ == "__main__":
    dv = torch.device("cuda:0")
    n, d, L, B = 32, 8, 3, 1024
    C, S, dd = n * (n - 1), 16 * n, 16
    g = torch.Generator().manual_seed(0)
    adjs = [((torch.rand(8, S, S, generator=g) - 0.45) * (2.0 / S ** 0.5)).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1) for _ in range(L)]
    h0 = torch.randn(8, C, S, 1, generator=g).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1, 1)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)

    def fwd():
        with torch.no_grad():
            propagate(adjs, h0, "relu", head, tail)
    for cfg in (sys.argv[1:] or ["c", "a"]):
        os.environ["RECON_BGEMM_CFG"] = cfg
        for abl in (0, 4, 1, 5, 2, 6):
            os.environ["RECON_BGEMM_ABL"] = str(abl)
            tf = _time(fwd, 3)
            print(json.dumps({"cfg": cfg, "ablate": abl, "what": {0: "full", 4: "no stores", 1: "copies from the zero page", 5: "zero page + no stores", 2: "no copies", 6: "no copies, no stores"}[abl],
                              "fwd_ms_incl_0.2_gather": tf * 1e3}), flush=True)
