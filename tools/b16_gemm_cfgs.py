"""A/B of the batched bf16 GEMM's tile shapes on the n = 32 propagation (forward = 3 NT products; backward = (c) + (d) per hop)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tools.secondary import _time
from recon_amd import _lib
from recon_amd.propagation import propagate, make_start_embedding, get_head_indices, get_tail_indices

if __name__ == "__main__":
    torch.autograd.set_multithreading_enabled(False)
    dv = torch.device("cuda:0")
    n, d, L, B = 32, 8, 3, 1024
    C, S, dd = n * (n - 1), 16 * n, 16
    g = torch.Generator().manual_seed(0)
    adjs = [((torch.rand(8, S, S, generator=g) - 0.45) * (2.0 / S ** 0.5)).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1).requires_grad_(True) for _ in range(L)]
    h0 = torch.randn(8, C, S, 1, generator=g).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1, 1).requires_grad_(True)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(dv)
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(dv)
    G = torch.randn(8, C, dd * L, generator=g).to(torch.bfloat16).to(dv).repeat(B // 8, 1, 1)

    def fwd():
        with torch.no_grad():
            propagate(adjs, h0.detach(), "relu", head, tail)

    def fwd_bwd():
        for t in adjs + [h0]:
            t.grad = None
        propagate(adjs, h0, "relu", head, tail).backward(G)
    ref = None
    for cfg in (sys.argv[1:] or ["a", "b", "c", "d", "e"]):
        _lib.config_set("RECON_BGEMM_CFG", cfg)
        with torch.no_grad():
            out = propagate(adjs, h0.detach(), "relu", head, tail)
        if ref is None:
            ref = out
        same = bool(torch.equal(out, ref))
        tf = _time(fwd, 3)
        tb = _time(fwd_bwd, 2)
        print(json.dumps({"cfg": cfg, "fwd_ms": tf * 1e3, "fwd_TFLOPs": 2.0 * B * S * S * C * L / tf / 1e12, "fwd_bwd_ms": tb * 1e3, "bit_equal_to_first": same}), flush=True)
