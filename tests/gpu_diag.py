#!/usr/bin/env python3
"""Stage-by-stage diagnosis of the GAT HIP path on the GPU box (prints max errors per stage; never
asserts).  Usage on the box:  python tests/gpu_diag.py   (a checker script, not collected by pytest)"""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import recon_oracle as O          # noqa: E402
from recon_amd import _lib                    # noqa: E402
from recon_amd.graph import GraphCSR          # noqa: E402
from recon_amd.gat_layers import _fwd_args    # noqa: E402


def err(a, b):
    a, b = a.detach().cpu().double(), b.detach().cpu().double()
    return "max|d|=%.3e (ref max %.3e) finite=%s" % ((a - b).abs().max().item() if a.numel() else 0.0,
                                                      b.abs().max().item() if b.numel() else 0.0,
                                                      bool(torch.isfinite(a).all()))


def main():
    d = torch.device("cuda:0")
    print(torch.cuda.get_device_name(0))
    L = _lib.lib()
    for (N, E, F_, R, D, H) in [(40, 160, 8, 8, 16, 2), (128, 512, 200, 200, 200, 2), (50, 200, 10, 6, 50, 1)]:
        print("== N=%d E=%d F=%d R=%d D=%d H=%d" % (N, E, F_, R, D, H))
        g = torch.Generator().manual_seed(1)
        edge = torch.randint(0, N, (2, E), generator=g)
        x = torch.randn(N, F_, generator=g)
        ee = torch.randn(E, R, generator=g)
        a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
        a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
        Gr = torch.randn(N, H * D, generator=g)
        G = GraphCSR(edge.to(d), N)
        torch.cuda.synchronize()
        order = torch.sort(edge[0], stable=True).indices
        print("eid ok:", bool((G.eid.cpu().long() == order).all()), " src ok:", bool((G.src.cpu().long() == edge[1][order]).all()))
        f32 = dict(dtype=torch.float32, device=d)
        xd, eed, ad, a2d = x.to(d), ee.to(d), a.to(d), a2.to(d)
        P = torch.full((2, H, N, D), float("nan"), **f32)
        Q = torch.full((H, E, D), float("nan"), **f32)
        sigma = torch.full((H, E), float("nan"), **f32)
        Z = torch.full((H, N), float("nan"), **f32)
        out = torch.full((N, H * D), float("nan"), **f32)
        args = _fwd_args(G, xd, eed, ad, a2d, None, P, Q, sigma, Z, out, 0.2, True)
        print("project rc", L.recon_gat_project(C.byref(G.c), C.byref(args), _lib.current_stream()))
        torch.cuda.synchronize()
        for h in range(H):
            A_dst, A_src, A_rel = a[h][:, :F_], a[h][:, F_:2 * F_], a[h][:, 2 * F_:]
            print(" h%d P_dst" % h, err(P[0, h], x @ A_dst.t()), "| P_src", err(P[1, h], x @ A_src.t()),
                  "| Q", err(Q[h], ee[order] @ A_rel.t()))
        print("edge_fwd rc", L.recon_gat_edge_fwd(C.byref(G.c), C.byref(args), _lib.current_stream()))
        torch.cuda.synchronize()
        Gm = torch.full((H, E, D), float("nan"), **f32)
        gP = torch.full((2, H, N, D), float("nan"), **f32)
        partial = torch.empty(L.recon_gat_bwd_partial_floats(N, E, F_, R, D, H), **f32)
        g_x = torch.full((N, F_), float("nan"), **f32)
        g_ee = torch.full((E, R), float("nan"), **f32)
        g_a = torch.full((H, D, 2 * F_ + R), float("nan"), **f32)
        g_a2 = torch.full((H, D), float("nan"), **f32)
        Grd = Gr.to(d)
        bargs = _lib.GatBwdArgs(args, Grd.data_ptr(), H * D, Gm.data_ptr(), gP.data_ptr(), partial.data_ptr(),
                                g_x.data_ptr(), g_ee.data_ptr(), g_a.data_ptr(), g_a2.data_ptr())
        print("bwd rc", L.recon_gat_bwd(C.byref(G.c), C.byref(bargs), _lib.current_stream()))
        torch.cuda.synchronize()
        gx_ref = torch.zeros(N, F_, dtype=torch.float64)
        gee_ref = torch.zeros(E, R, dtype=torch.float64)
        for h in range(H):
            r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, True,
                                     Gr[:, h * D:(h + 1) * D].double())
            print(" h%d out" % h, err(out[:, h * D:(h + 1) * D], r["out"]))
            print("    sigma", err(sigma[h], r["sigma"][order]), "| Z", err(Z[h], r["Z"]))
            print("    Gm", err(Gm[h], r["gm"][order]), "| gP_dst", err(gP[0, h], r["gP_dst"]), "| gP_src", err(gP[1, h], r["gP_src"]))
            print("    g_a", err(g_a[h], r["g_a"]), "| g_a2", err(g_a2[h:h + 1], r["g_a_2"]))
            gx_ref += r["g_x"]
            gee_ref += r["g_edge_embed"]
        print(" g_x", err(g_x, gx_ref), "| g_ee", err(g_ee, gee_ref))


if __name__ == "__main__":
    main()
