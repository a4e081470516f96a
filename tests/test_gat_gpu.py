"""GPU parity tests for the GAT hot path: HIP kernels (through the C ABI and the drop-in modules)
vs the CPU oracle and vs the golden vectors produced by the reference.  Tolerance for fp32 outputs:
1e-4 absolute (BASELINE.json north_star: parity within 1e-4 fp32), gradients 1e-4 relative to the
gradient's max magnitude."""
import ctypes as C

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import recon_oracle as O

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def dev():
    assert torch.cuda.is_available(), "these tests need an MI355X"
    return torch.device("cuda:0")


def close(actual, desired, atol=1e-4, rel_to_max=1e-4, what=""):
    actual = actual.detach().cpu().numpy() if torch.is_tensor(actual) else actual
    desired = desired.detach().cpu().numpy() if torch.is_tensor(desired) else desired
    tol = atol + rel_to_max * (np.abs(desired).max() if desired.size else 0.0)
    err = np.abs(actual - desired).max() if desired.size else 0.0
    assert np.isfinite(actual).all(), what + ": non-finite values"
    assert err <= tol, "%s: max abs err %.3e > tol %.3e" % (what, err, tol)


# ------------------------------------------------------------------------------- K3
@pytest.mark.parametrize("small", ["one_launch", "sort_chain"])
@pytest.mark.parametrize("N,E,seed", [(12, 40, 0), (1, 5, 1), (300, 5000, 2), (70000, 200000, 3), (50, 0, 4), (14541, 8000, 5), (32768, 8192, 6), (200, 8193, 7),
                                      (32769, 1000, 8), (3, 7000, 9)])
def test_graph_build(N, E, seed, small, recon_config):
    """Both builds: the one-launch kernel for small graphs (E <= 8 192, N <= 32 768) and the chain of radix-sort
    launches (RECON_GRAPH_SMALL=0 and everything larger): one-, two- and three-digit keys, hub rows, the limits themselves."""
    from recon_amd.graph import GraphCSR
    if small == "sort_chain":
        recon_config("RECON_GRAPH_SMALL", "0")
    g = torch.Generator().manual_seed(seed)
    edge = torch.randint(0, N, (2, E), generator=g)
    if E > 3000:
        edge[0, : E // 10] = N // 2                                    # a hub row
    G = GraphCSR(edge.to(dev()), N)
    torch.cuda.synchronize()
    order = torch.sort(edge[0], stable=True).indices
    np.testing.assert_array_equal(G.eid.cpu().numpy(), order.numpy().astype(np.int32))
    np.testing.assert_array_equal(G.dst.cpu().numpy(), edge[0][order].numpy())
    np.testing.assert_array_equal(G.src.cpu().numpy(), edge[1][order].numpy())
    counts = torch.bincount(edge[0], minlength=N)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    np.testing.assert_array_equal(G.rowptr_dst.cpu().numpy(), rowptr.numpy())
    src_sorted = edge[1][order]
    order2 = torch.sort(src_sorted, stable=True).indices
    np.testing.assert_array_equal(G.slot_by_src.cpu().numpy(), order2.numpy())
    counts = torch.bincount(edge[1], minlength=N)
    rowptr = torch.cat([torch.zeros(1, dtype=torch.long), counts.cumsum(0)])
    np.testing.assert_array_equal(G.rowptr_src.cpu().numpy(), rowptr.numpy())


def test_graph_build_rejects_out_of_range_ids():
    """The reference fails on an edge id outside [0, N) (index out of range); so does the drop-in, once per cached graph."""
    from recon_amd.graph import GraphCSR, prepare_graph
    edge = torch.tensor([[0, 1, 5], [1, 2, 0]], device=dev())
    with pytest.raises(IndexError):
        GraphCSR(edge, 5)
    with pytest.raises(IndexError):
        prepare_graph(torch.tensor([[0, 1], [1, -1]], device=dev()), None, 4)
    GraphCSR(edge, 6)


def test_graph_build_flags_out_of_range_ids_found_by_the_build():
    """More than HUB_CHUNK edges: the id check rides in the build and its flag comes back with the hub-table sizes — when something first
    needs the graph (GraphCSR.resolve), or at construction with the deferral off.  A clean graph between two bad ones is not blamed."""
    from recon_amd import graph as G
    gen = torch.Generator().manual_seed(5)
    N, E = 50, 300
    good = torch.randint(0, N, (2, E), generator=gen).to(dev())
    bad = good.clone()
    bad[1, 137] = N
    g_bad = G.GraphCSR(bad, N)
    g_ok = G.GraphCSR(good.clone(), N)
    assert g_bad.pending and g_ok.pending
    assert g_ok.n_hub >= 0 and not g_ok.pending                       # resolves; the shared flag is up, but these ids are fine
    for _ in range(2):                                                # the error is the graph's: every use raises it
        with pytest.raises(IndexError):
            g_bad.c
    assert G.GraphCSR(good.clone(), N).c.E == E
    G.DEFER_HUB_READ = False
    try:
        with pytest.raises(IndexError):
            G.GraphCSR(bad.clone(), N)
        assert not G.GraphCSR(good.clone(), N).pending
    finally:
        G.DEFER_HUB_READ = True
    # through the layer: the reference fails inside forward() too (index out of range)
    from recon_amd import gat_layers
    x = torch.randn(N, 8, generator=gen).to(dev()); ee = torch.randn(E, 4, generator=gen).to(dev())
    a = torch.randn(2, 6, 20, generator=gen).to(dev()); a2 = torch.randn(2, 6, generator=gen).to(dev())
    with pytest.raises(IndexError):
        gat_layers.gat_heads(x, ee, a, a2, G.prepare_graph(bad.clone(), None, N), None, 0.2, True)
    out = gat_layers.gat_heads(x, ee, a, a2, G.prepare_graph(good.clone(), None, N), None, 0.2, True)
    assert torch.isfinite(out).all()


def test_graph_deferred_hub_read_gives_the_same_layer_output():
    """gat_heads on a graph whose hub sizes are still on the device (score stage first, then the read) against the same call on the resolved graph."""
    from recon_amd import graph as G, gat_layers
    gen = torch.Generator().manual_seed(6)
    N, E = 40, 2000                                                    # in-degrees ~50 +- : some rows beyond HUB_CHUNK = 64
    edge = torch.stack((torch.randint(0, 8, (E,), generator=gen) * 5, torch.randint(0, N, (E,), generator=gen))).to(dev())
    x = torch.randn(N, 12, generator=gen).to(dev()); ee = torch.randn(E, 8, generator=gen).to(dev())
    a = torch.randn(3, 5, 32, generator=gen).to(dev()); a2 = torch.randn(3, 5, generator=gen).to(dev())
    g1 = G.GraphCSR(edge, N)
    assert g1.pending
    o1 = gat_layers.gat_heads(x, ee, a, a2, g1, None, 0.2, True)
    assert not g1.pending and g1.n_hub > 0
    o2 = gat_layers.gat_heads(x, ee, a, a2, g1, None, 0.2, True)
    assert torch.equal(o1, o2)


def test_graph_cache_distinguishes_strided_views():
    """Two views of one storage with equal data_ptr and shape but different strides are different edge lists."""
    from recon_amd.graph import prepare_graph
    base = torch.tensor([[0, 3, 1, 2, 2, 0, 3, 1], [1, 1, 0, 0, 3, 3, 2, 2]], device=dev())       # [2, 8]
    a = base[:, :4]                       # columns 0..3, stride (8, 1)
    b = base.view(-1)[:8].view(2, 4)      # same data_ptr and shape, stride (4, 1): rows are base[0,:4], base[0,4:]
    assert a.data_ptr() == b.data_ptr() and a.shape == b.shape and a.stride() != b.stride()
    ga, gb = prepare_graph(a, None, 4), prepare_graph(b, None, 4)
    assert ga is not gb
    np.testing.assert_array_equal(ga.src.cpu().numpy()[np.argsort(ga.eid.cpu().numpy())], a[1].cpu().numpy())
    np.testing.assert_array_equal(gb.src.cpu().numpy()[np.argsort(gb.eid.cpu().numpy())], b[1].cpu().numpy())


# ------------------------------------------------------------------------------- K4
@pytest.mark.parametrize("cfg", ["0", "1"])        # RECON_GEMM_CFG: default tile choice / 128x128 forced
@pytest.mark.parametrize("M,N,K,nk", [(128, 128, 16, 1), (200, 72, 50, 1), (33, 257, 19, 0), (1000, 400, 200, 0),
                                      (256, 1600, 200, 1), (700, 200, 600, 1), (513, 600, 200, 0), (300, 204, 37, 1)])
def test_sgemm(M, N, K, nk, cfg, recon_config):
    from recon_amd import _lib
    recon_config("RECON_GEMM_CFG", cfg)
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g)
    B = torch.randn(N, K, generator=g) if nk else torch.randn(K, N, generator=g)
    Ad, Bd = A.to(dev()), B.to(dev())
    Cd = torch.full((M, N), float("nan"), device=dev())
    rc = _lib.lib().recon_sgemm(M, N, K, Ad.data_ptr(), K, Bd.data_ptr(), Bd.shape[1], nk, Cd.data_ptr(), N,
                                _lib.current_stream())
    assert rc == 0
    ref = (A.double() @ (B.double().t() if nk else B.double())).float()
    close(Cd, ref, atol=1e-5, rel_to_max=2e-6, what="sgemm")


@pytest.mark.parametrize("M,N,K", [(128, 208, 32), (200, 72, 52), (33, 257, 20), (700, 200, 600), (513, 600, 200),
                                   (300, 204, 36), (1000, 25, 8)])
def test_sgemm_bx3(M, N, K):
    """Split-precision (3 x bf16) MFMA GEMM: fp32-class accuracy against an fp64 product."""
    from recon_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(2.0 * torch.randn(M, 1, generator=g))     # rows of very different scale
    B = torch.randn(N, K, generator=g)
    Ad, Bd = A.to(dev()), B.to(dev())
    Cd = torch.full((M, N), float("nan"), device=dev())
    L = _lib.lib()
    ws = torch.empty(L.recon_sgemm_bx3_workspace_bytes(N, K), dtype=torch.uint8, device=dev())
    rc = L.recon_sgemm_bx3(M, N, K, Ad.data_ptr(), K, Bd.data_ptr(), K, Cd.data_ptr(), N, ws.data_ptr(), _lib.current_stream())
    assert rc == 0
    ref = A.double() @ B.double().t()
    # elementwise bound of an fp32 dot product: |err| <= c * eps * sum_k |a_k b_k| with c = 16, eps = 2^-24
    # (the fp32 library GEMM reaches c = 6 on these inputs; a two-term bf16 split would need c = 256)
    bound = (A.double().abs() @ B.double().abs().t()) * (2.0 ** -20) + 1e-30
    err = (Cd.cpu().double() - ref).abs()
    assert torch.isfinite(Cd).all()
    assert (err <= bound).all(), float((err / bound).max())
    f32 = (A.to(dev()) @ B.to(dev()).t()).cpu().double()                 # the fp32 library product is no closer
    assert err.max() <= 4.0 * (f32 - ref).abs().max() + 1e-12


@pytest.mark.parametrize("M,N,K", [(128, 208, 64), (600, 200, 8192), (36, 216, 1000), (260, 24, 77), (4, 424, 33), (300, 300, 4096),
                                   (12, 5, 50)])
def test_sgemm_bx3_tn(M, N, K):
    """The k-major (weight-gradient) form of the split-precision GEMM: C = A^T B, A [K,M], B [K,N]."""
    from recon_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g) * torch.exp(2.0 * torch.randn(1, M, generator=g))
    B = torch.randn(K, N, generator=g)
    Ad, Bd = A.to(dev()), B.to(dev())
    Cd = torch.full((M, N), float("nan"), device=dev())
    L = _lib.lib()
    ws = torch.empty(L.recon_sgemm_bx3_tn_workspace_bytes(M, N, K), dtype=torch.uint8, device=dev())
    rc = L.recon_sgemm_bx3_tn(M, N, K, Ad.data_ptr(), M, Bd.data_ptr(), N, Cd.data_ptr(), N, ws.data_ptr(), _lib.current_stream())
    assert rc == 0
    ref = A.double().t() @ B.double()
    bound = (A.double().abs().t() @ B.double().abs()) * (2.0 ** -20) + 1e-30
    err = (Cd.cpu().double() - ref).abs()
    assert torch.isfinite(Cd).all()
    assert (err <= bound).all(), float((err / bound).max())


def _hx2_ws(nbytes):
    ws = torch.empty(nbytes + 256, dtype=torch.uint8, device=dev())
    off = (-ws.data_ptr()) % 256
    return ws, ws.data_ptr() + off


@pytest.mark.parametrize("scale", [1.0, 1e-6, 3e4])
@pytest.mark.parametrize("M,N,K", [(128, 208, 32), (200, 72, 56), (33, 257, 24), (700, 200, 600), (513, 600, 200),
                                   (300, 204, 40), (1000, 25, 8), (4100, 3300, 1032)])     # the last: the 256-row workgroup form (K >= 1024, >= 256 tiles)
def test_sgemm_hx2(M, N, K, scale):
    """Split-precision (2 x f16, per-tensor power-of-two scale) MFMA GEMM: fp32-class accuracy against an fp64 product, at
    magnitudes far outside half's own range."""
    from recon_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(M, K, generator=g) * torch.exp(2.0 * torch.randn(M, 1, generator=g)) * scale     # rows of very different scale
    B = torch.randn(N, K, generator=g) / scale
    Ad, Bd = A.to(dev()), B.to(dev())
    Cd = torch.full((M, N), float("nan"), device=dev())
    L = _lib.lib()
    ws, wsp = _hx2_ws(L.recon_sgemm_hx2_workspace_bytes(M, N, K))
    rc = L.recon_sgemm_hx2(M, N, K, Ad.data_ptr(), K, Bd.data_ptr(), K, Cd.data_ptr(), N, wsp, _lib.current_stream())
    assert rc == 0
    ref = A.double() @ B.double().t()
    # the bound test_sgemm_bx3 holds bf16 x 3 to, plus the floor of a per-TENSOR scale: elements more than 2^18 below their
    # tensor's maximum keep an absolute error of 2^-39 of that maximum (their low half term is subnormal)
    Aa, Ba = A.double().abs(), B.double().abs()
    bound = (Aa @ Ba.t()) * (2.0 ** -20) + (2.0 ** -38) * (Aa.max() * Ba.sum(1)[None, :] + Ba.max() * Aa.sum(1)[:, None]) + 1e-30
    err = (Cd.cpu().double() - ref).abs()
    assert torch.isfinite(Cd).all()
    assert (err <= bound).all(), float((err / bound).max())
    f32 = (Ad @ Bd.t()).cpu().double()
    assert err.max() <= 4.0 * (f32 - ref).abs().max() + 1e-12


@pytest.mark.parametrize("M,N,K", [(128, 208, 64), (600, 200, 8192), (36, 216, 1000), (260, 24, 77), (4, 424, 33), (300, 300, 4096),
                                   (12, 5, 50), (8, 8, 1)])
def test_sgemm_hx2_tn(M, N, K):
    """The k-major (weight-gradient) form of the 2 x f16 GEMM: C = A^T B, A [K,M], B [K,N]; K tails come from a page of zeros."""
    from recon_amd import _lib
    g = torch.Generator().manual_seed(M + N + K)
    A = torch.randn(K, M, generator=g) * torch.exp(2.0 * torch.randn(1, M, generator=g)) * 1e-3
    B = torch.randn(K, N, generator=g) * 50.0
    Ad, Bd = A.to(dev()), B.to(dev())
    Cd = torch.full((M, N), float("nan"), device=dev())
    L = _lib.lib()
    ws, wsp = _hx2_ws(L.recon_sgemm_hx2_tn_workspace_bytes(M, N, K))
    ws.fill_(0xFF)                                                     # NaN patterns everywhere the kernels must not read
    rc = L.recon_sgemm_hx2_tn(M, N, K, Ad.data_ptr(), M, Bd.data_ptr(), N, Cd.data_ptr(), N, wsp, _lib.current_stream())
    assert rc == 0
    ref = A.double().t() @ B.double()
    Aa, Ba = A.double().abs(), B.double().abs()
    bound = (Aa.t() @ Ba) * (2.0 ** -20) + (2.0 ** -38) * (Aa.max() * Ba.sum(0)[None, :] + Ba.max() * Aa.sum(0)[:, None]) + 1e-30
    err = (Cd.cpu().double() - ref).abs()
    assert torch.isfinite(Cd).all()
    assert (err <= bound).all(), float((err / bound).max())


# ------------------------------------------------------------------------------- G1-G3
@pytest.mark.parametrize("name", ["spmm1_o1", "spmm1_oD"])
def test_spmm_golden(name):
    from recon_amd.gat_layers import SpecialSpmmFinal
    g = load_golden(name)
    w = T(g["edge_w"]).to(dev()).requires_grad_(True)
    out = SpecialSpmmFinal()(T(g["edge"]).to(dev()), w, int(g["N"]), g["edge"].shape[1], g["edge_w"].shape[1])
    close(out, g["out"], atol=1e-6, what="spmm out")
    (out * T(g["G"]).to(dev())).sum().backward()
    np.testing.assert_array_equal(w.grad.cpu().numpy(), g["g_edge_w"])


# ------------------------------------------------------------------------------- G4
GAT_CASES = ["gat1_cfg1", "gat2_dups", "gat3_nhop", "gat4_noconcat", "gat5_train", "gat7_cfg2_slice"]


@pytest.mark.parametrize("path", ["auto", "proj"])
@pytest.mark.parametrize("name", GAT_CASES)
def test_gat_layer_golden(name, path, monkeypatch):
    """Drop-in SpGraphAttentionLayer vs the reference's own outputs / gradients, through both kernel
    formulations ('auto' picks aggregate-then-project where instantiated)."""
    from recon_amd import gat_layers
    from recon_amd.gat_layers import SpGraphAttentionLayer
    monkeypatch.setattr(gat_layers, "_GAT_PATH", path)
    g = load_golden(name)
    d = dev()
    N, F_ = g["x"].shape
    D = g["a"].shape[0]
    R = g["edge_embed"].shape[1]
    layer = SpGraphAttentionLayer(N, F_, D, R, dropout=float(g["train_p"]), alpha=float(g["alpha"]),
                                  concat=bool(g["concat"])).to(d)
    layer.load_state_dict({"a": T(g["a"]), "a_2": T(g["a_2"])}, strict=True)
    x = T(g["x"]).to(d).requires_grad_(True)
    ee = T(g["edge_embed"]).to(d).requires_grad_(True)
    has = "edge_nhop" in g
    nhop = T(g["edge_nhop"]).to(d) if has else torch.tensor([])
    ee2 = T(g["edge_embed_nhop"]).to(d).requires_grad_(True) if has else torch.tensor([])
    if float(g["train_p"]) > 0:
        layer.train()
        mask = T(g["mask"]).to(d)
        layer.draw_keep = lambda E, device: mask.view(1, E)        # replay the reference's recorded mask
    else:
        layer.eval()
    out = layer(x, T(g["edge"]).to(d), ee, nhop, ee2)
    close(out, g["out"], what=name + " out")
    (out * T(g["G"]).to(d)).sum().backward()
    close(x.grad, g["g_x"], atol=1e-5, what=name + " g_x")
    close(ee.grad, g["g_edge_embed"], atol=1e-5, what=name + " g_edge_embed")
    close(layer.a.grad, g["g_a"], atol=1e-5, what=name + " g_a")
    close(layer.a_2.grad, g["g_a_2"], atol=1e-5, what=name + " g_a_2")
    if has:
        close(ee2.grad, g["g_edge_embed_nhop"], atol=1e-5, what=name + " g_edge_embed_nhop")


@pytest.mark.parametrize("N,E,F_,R,D,H,concat", [
    (64, 300, 16, 8, 32, 3, True),        # VEC4, G=8
    (50, 200, 10, 6, 50, 2, True),        # VEC2
    (40, 160, 7, 5, 25, 4, False),        # VEC1 (odd D)
    (100, 450, 12, 12, 100, 2, True),     # G=32
    (30, 120, 24, 16, 400, 2, True),      # KR=2
    (20, 90, 48, 48, 1600, 1, False),     # out_att-sized head: KR=8
    (16, 0, 8, 8, 16, 2, True),           # no edges at all
    (24, 100, 1600, 1600, 40, 1, False),  # out_att-sized inputs (F = R = 1600)
    (36, 140, 12, 8, 20, 11, True),       # 11 heads: two head groups walked by one wave
    (30, 120, 264, 300, 24, 2, True),     # F, R > 256: two register rows per lane
    (12, 40, 1040, 64, 16, 8, True),      # 8 heads x (2F+R) floats of score vectors exceed the 64 KiB LDS stage: falls back to 'proj'
])
@pytest.mark.parametrize("path", ["atp", "proj"])
def test_gat_heads_vs_oracle(N, E, F_, R, D, H, concat, path, monkeypatch):
    """Fused H-head call (with dropout factors): outputs and all gradients vs the oracle, both formulations."""
    from recon_amd import gat_layers
    from recon_amd.gat_layers import gat_heads
    from recon_amd.graph import prepare_graph
    monkeypatch.setattr(gat_layers, "_GAT_PATH", path)
    d = dev()
    g = torch.Generator().manual_seed(N * 7 + D)
    edge = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, F_, generator=g)
    ee = torch.randn(E, R, generator=g)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    keep = (torch.rand(H, E, generator=g) > 0.3).float() / 0.7
    G = torch.randn(N, H * D, generator=g)
    xd, eed, ad, a2d = (t.to(d).requires_grad_(True) for t in (x, ee, a, a2))
    graph = prepare_graph(edge.to(d), None, N)
    out = gat_heads(xd, eed, ad, a2d, graph, keep.to(d), 0.2, concat)
    (out * G.to(d)).sum().backward()
    g_x = torch.zeros_like(x)
    g_ee = torch.zeros_like(ee)
    for h in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(),
                                 0.2, concat, G[:, h * D:(h + 1) * D].double(), mask=keep[h].double())
        close(out[:, h * D:(h + 1) * D], r["out"].float(), what="out h%d" % h)
        close(ad.grad[h], r["g_a"].float(), atol=1e-5, what="g_a h%d" % h)
        close(a2d.grad[h:h + 1], r["g_a_2"].float(), atol=1e-5, what="g_a_2 h%d" % h)
        g_x += r["g_x"].float()
        g_ee += r["g_edge_embed"].float()
    close(xd.grad, g_x, atol=1e-5, what="g_x")
    close(eed.grad, g_ee, atol=1e-5, what="g_edge_embed")


def test_gat_eval_no_grad_and_determinism():
    from recon_amd.gat_layers import SpGraphAttentionLayer
    d = dev()
    x, edge, ee = O.synthetic_batched_graph(16, 16, 64, 32, 32, seed=1)
    torch.manual_seed(0)
    layer = SpGraphAttentionLayer(256, 32, 64, 32, dropout=0.3, alpha=0.2).to(d).eval()
    with torch.no_grad():
        y1 = layer(x.to(d), edge.to(d), ee.to(d), torch.tensor([]), torch.tensor([]))
        y2 = layer(x.to(d), edge.to(d), ee.to(d), torch.tensor([]), torch.tensor([]))
    assert torch.equal(y1, y2)                       # fixed summation order: bitwise reproducible
    ref = O.gat_layer_forward(x, edge, ee, None, None, layer.a.detach().cpu(), layer.a_2.detach().cpu(), 0.2, True)
    close(y1, ref, what="eval forward")


@pytest.mark.parametrize("name", ["spgat1_nhop", "spgat2_1hop"])
def test_spgat_golden(name):
    """Fused-heads SpGAT harness vs the reference SpGAT (loads the reference's state_dict strictly)."""
    from recon_amd.models import SpGAT
    g = load_golden(name)
    d = dev()
    H, nhid = int(g["nheads"]), int(g["nhid"])
    N, nfeat = g["x"].shape
    m = SpGAT(N, nfeat, nhid, g["rel"].shape[1], dropout=0.0, alpha=float(g["alpha"]), nheads=H).to(d)
    m.load_state_dict({k[2:]: T(v) for k, v in g.items() if k.startswith("p.")}, strict=True)
    m.eval()
    x = T(g["x"]).to(d).requires_grad_(True)
    rel = T(g["rel"]).to(d).requires_grad_(True)
    has = "edge_nhop" in g
    nhop = T(g["edge_nhop"]).to(d) if has else torch.tensor([])
    ntype = T(g["edge_type_nhop"]).to(d) if has else torch.tensor([])
    et = T(g["edge_type"]).to(d)
    out, out_rel = m(None, x, rel, T(g["edge"]).to(d), et, rel[et], nhop, ntype)
    close(out, g["out"], what="spgat out")
    close(out_rel, g["out_rel"], what="spgat out_rel")
    ((out * T(g["G"]).to(d)).sum() + (out_rel * T(g["G2"]).to(d)).sum()).backward()
    close(x.grad, g["g_x"], atol=1e-5, what="g_x")
    close(rel.grad, g["g_rel"], atol=1e-5, what="g_rel")
    for k, p in m.named_parameters():
        close(p.grad, g["g." + k], atol=1e-5, what="g." + k)


def test_full_size_cfg2_properties():
    """BASELINE.json configs[1] at full size (B=512, n=16, 64 e/graph, F=R=D=200, H=8), size-independent properties:
    (i) graphs of a batch are independent (a slice against the oracle), (ii) linearity of the backward in grad_out,
    (iii) permutation invariance: shuffling the edge columns changes nothing but fp order.  Every output and gradient of
    this configuration against the float64 oracle: tests/test_full_size_gpu.py."""
    from recon_amd.gat_layers import gat_heads
    from recon_amd.graph import prepare_graph
    d = dev()
    B, n, e, F_, R, D, H = 512, 16, 64, 200, 200, 200, 8
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, R, seed=0)
    g = torch.Generator().manual_seed(0)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    xd, eed, ad, a2d = (t.to(d) for t in (x, ee, a, a2))
    edged = edge.to(d)
    graph = prepare_graph(edged, None, B * n)
    xd.requires_grad_(True)
    out = gat_heads(xd, eed, ad, a2d, graph, None, 0.2, True)
    # (i) graphs 0..3 are independent of the rest of the batch
    nb, eb = 4 * n, 4 * e
    for h in (0, 5):
        ref = O.gat_layer_forward(x[:nb], edge[:, :eb], ee[:eb], None, None, a[h], a2[h:h + 1], 0.2, True)
        close(out[:nb, h * D:(h + 1) * D], ref, what="cfg2 slice head %d" % h)
    # (ii) linearity of the backward
    G1 = torch.randn(out.shape, generator=g).to(d)
    G2 = torch.randn(out.shape, generator=g).to(d)
    g1, = torch.autograd.grad(out, xd, G1, retain_graph=True)
    g2, = torch.autograd.grad(out, xd, G2, retain_graph=True)
    g12, = torch.autograd.grad(out, xd, G1 + 2 * G2)
    close(g12, g1 + 2 * g2, atol=1e-4, what="backward linearity")
    # (iii) edge order invariance
    perm = torch.randperm(edge.shape[1], generator=g).to(d)
    graph2 = prepare_graph(edged[:, perm].contiguous(), None, B * n)
    out2 = gat_heads(xd.detach(), eed[perm].contiguous(), ad, a2d, graph2, None, 0.2, True)
    close(out2, out, atol=2e-5, what="edge permutation invariance")


@pytest.mark.parametrize("family", ["2", "1", "0"])
def test_index_range_beyond_2_31_elements(family, monkeypatch):
    """Maximum sizes: N = 3.2 M nodes x 4 heads x (2F + R = 192) is 2.46 G elements of V / g_V (9.8 GB each) — every row offset past
    node 2.8 M needs 64-bit arithmetic.  The graph is TWO copies of one half (same features, node ids shifted by N/2), so the second
    copy's outputs and input gradients must repeat the first's (same per-node work, only the addresses differ), the weight gradients
    are sums over both, and a sample of the first copy's nodes is checked against the oracle on their induced sub-problem."""
    from recon_amd import gat_layers
    from recon_amd.gat_layers import gat_heads
    from recon_amd.graph import prepare_graph, clear_graph_cache
    monkeypatch.setattr(gat_layers, "_GEMM_BX3", family)                 # every GEMM family has its own offset arithmetic
    d = dev()
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~60 GB of device memory")
    Nh, Eh, F_, R, D, H = 1600000, 2400000, 64, 64, 32, 4
    g = torch.Generator().manual_seed(11)
    dst = torch.randint(0, Nh, (Eh,), generator=g); src = torch.randint(0, Nh, (Eh,), generator=g)
    dst[:3000] = 7                                                    # one hub row (pieces) in each copy
    xh = torch.randn(Nh, F_, generator=g); eeh = torch.randn(Eh, R, generator=g) * 0.5
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    Gh = torch.randn(Nh, H * D, generator=g)
    edge = torch.stack([torch.cat([dst, dst + Nh]), torch.cat([src, src + Nh])])
    N = 2 * Nh
    assert N * H * (2 * F_ + R) > 2 ** 31
    xd = torch.cat([xh, xh]).to(d).requires_grad_(True)
    eed = torch.cat([eeh, eeh]).to(d).requires_grad_(True)
    ad, a2d = a.to(d).requires_grad_(True), a2.to(d).requires_grad_(True)
    clear_graph_cache()
    graph = prepare_graph(edge.to(d), None, N)
    assert graph.n_hub == 2
    out = gat_heads(xd, eed, ad, a2d, graph, None, 0.2, True)
    Gd = torch.cat([Gh, Gh]).to(d)
    out.backward(Gd)
    assert torch.equal(out[:Nh], out[Nh:]), "second copy of the graph (offsets beyond 2^31 elements) differs from the first"
    close(xd.grad[Nh:], xd.grad[:Nh], atol=1e-6, rel_to_max=1e-6, what="g_x of the second copy")
    close(eed.grad[Eh:], eed.grad[:Eh], atol=1e-6, rel_to_max=1e-6, what="g_edge_embed of the second copy")
    # a sample of destination nodes of the first copy against the oracle on their in-edges (forward) ...
    sample = torch.cat([torch.tensor([7]), torch.randint(0, Nh, (300,), generator=g)]).unique()
    emask = torch.isin(dst, sample)
    nodes = torch.cat([sample, src[emask]]).unique()
    relabel = torch.full((Nh,), -1, dtype=torch.long); relabel[nodes] = torch.arange(nodes.numel())
    sub_edge = torch.stack([relabel[dst[emask]], relabel[src[emask]]])
    outc = out[:Nh].detach().cpu()
    for h in range(H):
        ref = O.gat_layer_forward(xh[nodes], sub_edge, eeh[emask], None, None, a[h], a2[h:h + 1], 0.2, True)
        close(outc[sample][:, h * D:(h + 1) * D], ref[relabel[sample]], what="sampled rows, head %d" % h)
    # ... and the weight gradient against a second run on ONE copy: sums over both copies = 2 x
    del out, Gd, graph
    clear_graph_cache()
    x1 = xh.to(d); e1 = eeh.to(d)
    a1, a21 = a.to(d).requires_grad_(True), a2.to(d).requires_grad_(True)
    out1 = gat_heads(x1, e1, a1, a21, prepare_graph(torch.stack([dst, src]).to(d), None, Nh), None, 0.2, True)
    out1.backward(Gh.to(d))
    close(ad.grad, 2 * a1.grad, atol=1e-3, rel_to_max=2e-5, what="g_a over both copies")
    close(a2d.grad, 2 * a21.grad, atol=1e-3, rel_to_max=2e-5, what="g_a_2 over both copies")
    clear_graph_cache()


@pytest.mark.parametrize("family", ["2", "1"])
def test_full_size_cfg2_split_precision_vs_fp32_gemm(family, monkeypatch):
    """BASELINE.json configs[1] at full size: the layer on the split-precision GEMMs (2 half terms under a per-tensor scale,
    or 3 bf16 terms per fp32 operand) and on the exact-fp32 MFMA GEMMs must agree in outputs and in every gradient to fp32
    round-off."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    d = dev()
    B, n, e, F_, R, D, H = 512, 16, 64, 200, 200, 200, 8
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, R, seed=0)
    g = torch.Generator().manual_seed(0)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    G = torch.randn(B * n, H * D, generator=g).to(d)
    graph = prepare_graph(edge.to(d), None, B * n)
    res = {}
    for mode in (family, "0"):
        monkeypatch.setattr(gat_layers, "_GEMM_BX3", mode)
        leaves = [t.to(d).requires_grad_(True) for t in (x, ee, a, a2)]
        out = gat_layers.gat_heads(*leaves, graph, None, 0.2, True)
        grads = torch.autograd.grad(out, leaves, G)
        res[mode] = [out.detach()] + [t.detach() for t in grads]
    for name, u, v in zip(("out", "g_x", "g_edge_embed", "g_a", "g_a_2"), res[family], res["0"]):
        close(u, v, atol=2e-5, rel_to_max=4e-6, what="bx3 vs fp32 GEMM: " + name)   # 4e-6 of the largest element: a few ulp of the big sums


@pytest.mark.parametrize("path", ["atp", "proj"])
@pytest.mark.parametrize("N,E,F_,R,D,H,concat", [
    (64, 300, 16, 8, 32, 3, True),        # H = 3 -> head tile 4 with one masked head
    (50, 200, 10, 6, 50, 2, True),        # vector width 2
    (100, 450, 12, 12, 100, 8, True),     # 8 heads per wave
    (30, 120, 264, 300, 40, 2, False),    # two register rows per lane (F, R > 256)
    (20, 90, 1600, 1600, 64, 1, False),   # out_att-sized inputs
    (40, 150, 8, 8, 16, 11, True),        # 11 heads -> two head groups
    (16, 0, 8, 8, 16, 2, True),           # no edges
])
def test_gat_inference_both_formulations(path, N, E, F_, R, D, H, concat, monkeypatch):
    """No-grad forward through the aggregate-then-project kernels and through the project-then-aggregate
    kernels: both must match the oracle (they differ in fp32 summation order only)."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    monkeypatch.setattr(gat_layers, "_GAT_PATH", path)
    d = dev()
    g = torch.Generator().manual_seed(N * 3 + H)
    edge = torch.randint(0, N, (2, E), generator=g)
    x = torch.randn(N, F_, generator=g)
    ee = torch.randn(E, R, generator=g)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    graph = prepare_graph(edge.to(d), None, N)
    with torch.no_grad():
        out = gat_layers.gat_heads(x.to(d), ee.to(d), a.to(d), a2.to(d), graph, None, 0.2, concat)
    for h in range(H):
        ref = O.gat_layer_forward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, concat)
        close(out[:, h * D:(h + 1) * D], ref.float(), what="%s head %d" % (path, h))


def _power_law_batch(n_graphs, seed=0, max_n=256):
    """SURVEY 8d cfg-5 generator: n ~ U{16..256}, e = min(4096, 16 n), dst ~ Zipf(1) over the graph's nodes."""
    rs = np.random.RandomState(seed)
    dsts, srcs, base = [], [], 0
    for _ in range(n_graphs):
        n = int(rs.randint(16, max_n + 1))
        e = min(4096, 16 * n)
        p = 1.0 / np.arange(1, n + 1)
        p /= p.sum()
        dsts.append(rs.choice(n, size=e, p=p) + base)
        srcs.append(rs.randint(0, n, size=e) + base)
        base += n
    edge = torch.from_numpy(np.stack([np.concatenate(dsts), np.concatenate(srcs)])).long()
    return edge, base


@pytest.mark.parametrize("path", ["atp", "proj"])
def test_power_law_graphs(path, monkeypatch):
    """cfg-5-like skewed degree distribution (hub destinations with > 1000 edges, many isolated nodes)."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    monkeypatch.setattr(gat_layers, "_GAT_PATH", path)
    d = dev()
    edge, N = _power_law_batch(6, seed=1)
    E = edge.shape[1]
    F_, R, D, H = 24, 16, 32, 4
    g = torch.Generator().manual_seed(0)
    x = torch.randn(N, F_, generator=g)
    ee = torch.randn(E, R, generator=g) * 0.5
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    G = torch.randn(N, H * D, generator=g)
    deg = torch.bincount(edge[0], minlength=N)
    assert deg.max() > 500 and (deg == 0).sum() > 0
    xd, eed, ad, a2d = (t.to(d).requires_grad_(True) for t in (x, ee, a, a2))
    out = gat_layers.gat_heads(xd, eed, ad, a2d, prepare_graph(edge.to(d), None, N), None, 0.2, True)
    (out * G.to(d)).sum().backward()
    gx = torch.zeros_like(x)
    for h in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, True,
                                 G[:, h * D:(h + 1) * D].double())
        close(out[:, h * D:(h + 1) * D], r["out"].float(), what="power-law out h%d" % h)
        close(ad.grad[h], r["g_a"].float(), atol=1e-4, what="g_a h%d" % h)
        gx += r["g_x"].float()
    close(xd.grad, gx, atol=1e-4, what="g_x")


@pytest.mark.parametrize("M,K,N", [(237, 100, 200), (64, 200, 1600), (3, 5, 7), (1, 1, 1), (500, 33, 129), (4, 8, 4), (20, 36, 52), (237, 1600, 200)])
def test_small_mm(M, K, N):
    """The models' small dense products (relation_embed.mm(W)) on recon_sgemm: values and both gradients against float64."""
    from recon_amd.gat_layers import small_mm, _SmallMM
    d = dev()
    g = torch.Generator().manual_seed(M + N)
    A, B, G = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd = A.to(d).requires_grad_(True), B.to(d).requires_grad_(True)
    out = small_mm(Ad, Bd)
    assert isinstance(out.grad_fn, _SmallMM._backward_cls) or out.grad_fn.name().startswith("_SmallMM")
    (out * G.to(d)).sum().backward()
    close(out, (A.double() @ B.double()).float(), atol=1e-5, rel_to_max=1e-5, what="A B")
    close(Ad.grad, (G.double() @ B.double().t()).float(), atol=1e-5, rel_to_max=1e-5, what="g_A")
    close(Bd.grad, (A.double().t() @ G.double()).float(), atol=1e-5, rel_to_max=1e-5, what="g_B")
    big = small_mm(torch.randn(4096, 256, device=d), torch.randn(256, 512, device=d, requires_grad=True))      # above the threshold: torch.mm
    assert not big.grad_fn.name().startswith("_SmallMM")


@pytest.mark.parametrize("rows,K,N", [(14541, 50, 200), (5000, 33, 70), (2048, 8, 4), (20000, 256, 256)])
def test_thin_weight_mm(rows, K, N):
    """`entity_embeddings.mm(W_entities)` (GAT/models.py:177): the weight gradient A^T g on recon_sgemm_small with the long dimension cut
    over workgroups; values and both gradients against float64."""
    from recon_amd.gat_layers import small_mm
    d = dev()
    g = torch.Generator().manual_seed(rows + N)
    A, B, G = torch.randn(rows, K, generator=g), torch.randn(K, N, generator=g), torch.randn(rows, N, generator=g)
    Ad, Bd = A.to(d).requires_grad_(True), B.to(d).requires_grad_(True)
    out = small_mm(Ad, Bd)
    assert out.grad_fn.name().startswith(("_ThinWeightMM", "_SmallMM"))          # the smallest shape is a "small" product outright
    (out * G.to(d)).sum().backward()
    close(out, (A.double() @ B.double()).float(), atol=1e-4, rel_to_max=1e-5, what="A B")
    close(Ad.grad, (G.double() @ B.double().t()).float(), atol=1e-4, rel_to_max=1e-5, what="g_A")
    close(Bd.grad, (A.double().t() @ G.double()).float(), atol=1e-4, rel_to_max=2e-6, what="g_B")


@pytest.mark.parametrize("M,K,N", [(3000, 700, 900), (14541, 200, 400), (513, 1000, 77), (100, 4096, 2000), (1, 5000, 3000)])
def test_large_mm_runs_on_the_library_free_path(M, K, N):
    """small_mm above its small / thin-weight classes: this library's fp32 matrix-core GEMM (recon_sgemm_ex), never torch.mm — values and
    both gradients against float64 at fp32 round-off of a K-term (gradients: M- / N-term) dot product."""
    from recon_amd.gat_layers import small_mm
    d = dev()
    g = torch.Generator().manual_seed(M + N)
    A, B, G = torch.randn(M, K, generator=g), torch.randn(K, N, generator=g), torch.randn(M, N, generator=g)
    Ad, Bd = A.to(d).requires_grad_(True), B.to(d).requires_grad_(True)
    out = small_mm(Ad, Bd)
    assert out.grad_fn.name().startswith(("_GemmMM", "_ThinWeightMM", "_SmallMM"))
    (out * G.to(d)).sum().backward()
    close(out, (A.double() @ B.double()).float(), atol=1e-4, rel_to_max=1e-5, what="A B")
    close(Ad.grad, (G.double() @ B.double().t()).float(), atol=1e-4, rel_to_max=1e-5, what="g_A")
    close(Bd.grad, (A.double().t() @ G.double()).float(), atol=1e-4, rel_to_max=1e-5, what="g_B")


def _hub_graph(N, degs, seed, src_hubs=()):
    """Destination i gets degs[i] in-edges (0 for i >= len(degs)), sources uniform except that node j is the source of exactly c
    edges for every (j, c) in src_hubs; columns shuffled."""
    rs = np.random.RandomState(seed)
    dst = np.repeat(np.arange(len(degs)), degs)
    others = np.setdiff1d(np.arange(N), [j for j, _ in src_hubs])
    src = others[rs.randint(0, others.size, size=dst.size)]
    at = 0
    order = rs.permutation(dst.size)
    for j, c in src_hubs:
        src[order[at:at + c]] = j
        at += c
    perm = rs.permutation(dst.size)
    return torch.from_numpy(np.stack([dst[perm], src[perm]])).long()


def test_hub_tables():
    """recon_graph_hubs_count / _fill against numpy: rows longer than 64 slots, in node order, in pieces of <= 64 slots."""
    from recon_amd.graph import GraphCSR, HUB_CHUNK
    d = dev()
    rs = np.random.RandomState(5)
    N = 5000
    degs = rs.randint(0, 40, size=N)
    hubs = rs.choice(N, size=37, replace=False)
    degs[hubs] = rs.randint(65, 900, size=37)
    degs[hubs[0]] = 65; degs[hubs[1]] = 128; degs[hubs[2]] = 129; degs[N - 1] = 300; degs[0] = 64
    edge = _hub_graph(N, degs, 1, src_hubs=((17, 65), (4000, 1000), (N - 1, 64), (3, 129)))
    g = GraphCSR(edge.to(d), N)

    def tables(deg):
        want = np.nonzero(deg > HUB_CHUNK)[0]
        rowptr = np.concatenate([[0], np.cumsum(deg)])
        pieces, ptr = [], [0]
        for t, i in enumerate(want):
            for b in range(rowptr[i], rowptr[i + 1], HUB_CHUNK):
                pieces.append((i, b, min(b + HUB_CHUNK, rowptr[i + 1]), t))
            ptr.append(len(pieces))
        return want, np.array(ptr), np.array(pieces)

    want, ptr, pieces = tables(degs)
    assert g.n_hub == want.size and g.n_piece == len(pieces)
    assert np.array_equal(g.hub_node.cpu().numpy(), want)
    assert np.array_equal(g.hub_ptr.cpu().numpy(), ptr)
    assert np.array_equal(g.piece.cpu().numpy(), pieces)
    want, ptr, pieces = tables(np.bincount(edge[1].numpy(), minlength=N))
    assert list(want) == [3, 17, 4000] and g.n_hub_src == 3 and g.n_piece_src == len(pieces)
    assert np.array_equal(g.hub_node_src.cpu().numpy(), want)
    assert np.array_equal(g.hub_ptr_src.cpu().numpy(), ptr)
    assert np.array_equal(g.piece_src.cpu().numpy(), pieces)
    dst = np.repeat(np.arange(100), 64)                                  # every row of both views exactly 64 long: no hub
    g0 = GraphCSR(torch.from_numpy(np.stack([dst, (dst + 1 + np.arange(6400) % 64) % 100])).long().to(d), 100)
    assert g0.n_hub == 0 and g0.n_hub_src == 0


@pytest.mark.parametrize("N,F_,R,D,H,concat,drop,train", [
    (90, 24, 16, 32, 4, True, False, True),       # f16 x 2 capable: half-term V rows, destination part shared
    (90, 24, 16, 32, 4, True, True, True),        # attention dropout: Zk != Z, destination part per head
    (60, 10, 6, 50, 2, True, False, True),        # VEC 2
    (70, 12, 8, 20, 11, True, True, True),        # 11 heads: two head groups by one wave, per piece
    (40, 264, 300, 24, 2, True, False, True),     # two register rows per lane
    (50, 200, 200, 40, 1, False, False, True),    # out_att-like single head
    (90, 24, 16, 32, 4, True, False, False),      # inference: no Z / sigma
])
def test_hub_rows_split_vs_oracle_and_unsplit(N, F_, R, D, H, concat, drop, train, monkeypatch):
    """Destination rows of 65 ... 700 slots walked in pieces by several wavefronts: outputs and all gradients against the oracle
    and against the one-wave-per-row walk of the same kernels (HUB_CHUNK = 0)."""
    from recon_amd import gat_layers, graph as graph_mod
    monkeypatch.setattr(gat_layers, "_GAT_PATH", "atp")
    d = dev()
    degs = np.zeros(N, dtype=np.int64)
    degs[:12] = [700, 65, 64, 128, 129, 3, 0, 191, 66, 1, 320, 5]
    degs[12:N - 3] = np.random.RandomState(N).randint(0, 9, size=N - 15)
    degs[N - 1] = 100                                              # the last node a hub
    edge = _hub_graph(N, degs, 3, src_hubs=((N - 2, 300), (7, 65), (0, 130)))        # and three source-side hubs (CSC walk of the backward)
    E = edge.shape[1]
    g = torch.Generator().manual_seed(N + D)
    x = torch.randn(N, F_, generator=g)
    ee = torch.randn(E, R, generator=g) * 0.5
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    keep = (torch.rand(H, E, generator=g) > 0.3).float() / 0.7 if drop else None
    G = torch.randn(N, (H * D) if concat else D * H, generator=g)
    res = {}
    for chunk in (graph_mod.HUB_CHUNK, 0):
        monkeypatch.setattr(graph_mod, "HUB_CHUNK", chunk)
        graph_mod.clear_graph_cache()
        gr = graph_mod.prepare_graph(edge.to(d), None, N)
        assert (gr.n_hub == 8 and gr.n_hub_src >= 3) if chunk else (gr.n_hub == 0 and gr.n_hub_src == 0)
        xd, eed, ad, a2d = (t.to(d).requires_grad_(train) for t in (x, ee, a, a2))
        kd = keep.to(d) if drop else None
        if train:
            out = gat_layers.gat_heads(xd, eed, ad, a2d, gr, kd, 0.2, concat)
            (out * G.to(d)).sum().backward()
            res[chunk] = [t.detach().cpu() for t in (out, xd.grad, eed.grad, ad.grad, a2d.grad)]
        else:
            with torch.no_grad():
                res[chunk] = [gat_layers.gat_heads(xd, eed, ad, a2d, gr, None, 0.2, concat).cpu()]
    graph_mod.clear_graph_cache()
    names = ("out", "g_x", "g_edge_embed", "g_a", "g_a_2")
    for nm, s_, u_ in zip(names, res[graph_mod.HUB_CHUNK], res[0]):
        close(s_, u_, atol=2e-5, rel_to_max=2e-5, what="split vs unsplit " + nm)
    out, rest = res[graph_mod.HUB_CHUNK][0], res[graph_mod.HUB_CHUNK][1:]
    g_x = torch.zeros_like(x); g_ee = torch.zeros_like(ee)
    for h in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, concat,
                                 G[:, h * D:(h + 1) * D].double(), mask=keep[h].double() if drop else None)
        close(out[:, h * D:(h + 1) * D], r["out"].float(), what="out h%d" % h)
        if train:
            close(rest[2][h], r["g_a"].float(), atol=1e-4, what="g_a h%d" % h)
            close(rest[3][h:h + 1], r["g_a_2"].float(), atol=1e-4, what="g_a_2 h%d" % h)
            g_x += r["g_x"].float(); g_ee += r["g_edge_embed"].float()
    if train:
        close(rest[0], g_x, atol=1e-4, what="g_x")
        close(rest[1], g_ee, atol=1e-4, what="g_edge_embed")


@pytest.mark.parametrize("N,F_,R,D,H,concat,drop,train,table", [
    (900, 24, 16, 32, 4, True, False, True, False),      # f16 x 2 capable: half-term V rows, destination part shared
    (900, 24, 16, 32, 4, True, True, True, False),       # attention dropout
    (600, 10, 6, 50, 2, True, False, True, False),       # VEC 2, fp32 GEMMs
    (700, 12, 8, 20, 11, True, True, True, False),       # 11 heads
    (500, 200, 200, 40, 1, False, False, True, False),   # out_att-like single head, no ELU inside (grad_out goes into the GEMM as it is)
    (14541, 50, 50, 100, 2, True, True, True, True),     # the stage-A first layer: 128 live rows of 14 541, relation table
    (900, 24, 16, 32, 4, True, False, False, False),     # inference
])
def test_rows_with_edges_compacted_vs_oracle_and_uncompacted(N, F_, R, D, H, concat, drop, train, table, monkeypatch):
    """Few destination rows have edges (a knowledge-graph batch: GAT/main.py:478-516): the layer runs over those rows only (recon_graph.n_rows).
    Outputs and all gradients against the same kernels over every node (ROWS_COMPACT_MAX = 0) and against the oracle; rows without edges are
    exactly zero (GAT/layers.py:152-158)."""
    from recon_amd import gat_layers, graph as graph_mod
    monkeypatch.setattr(gat_layers, "_GAT_PATH", "atp")
    d = dev()
    rs = np.random.RandomState(N + H)
    live = np.sort(rs.choice(N, size=min(128, N // 6), replace=False))
    live[-1] = N - 1; live[0] = 0                                    # the first and the last node among them
    degs = np.zeros(N, dtype=np.int64)
    degs[live] = rs.randint(1, 30, size=live.size)
    degs[live[:6]] = [700, 65, 64, 128, 129, 300]                    # hub rows: the pieces name rows
    edge = _hub_graph(N, degs, 3, src_hubs=((N - 2, 300), (7, 65)))
    E = edge.shape[1]
    g = torch.Generator().manual_seed(N + D)
    x = torch.randn(N, F_, generator=g)
    nrel = 37
    ee = torch.randn(nrel if table else E, R, generator=g) * 0.5
    ee_index = torch.randint(0, nrel, (E,), generator=g) if table else None
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    keep = (torch.rand(H, E, generator=g) > 0.3).float() / 0.7 if drop else None
    G = torch.randn(N, H * D, generator=g)
    res = {}
    for frac in (0.7, 0.0):
        monkeypatch.setattr(graph_mod, "ROWS_COMPACT_MAX", frac)
        graph_mod.clear_graph_cache()
        gr = graph_mod.prepare_graph(edge.to(d), None, N)
        assert gr.n_rows == (live.size if frac else 0) and gr.n_hub == 5
        if frac:
            rows = gr._rows.cpu().numpy()
            assert np.array_equal(rows[:live.size], live)
            assert np.array_equal(rows[live.size:2 * live.size + 1], np.concatenate([[0], np.cumsum(degs[live])]))
            node_row = np.full(N, -1); node_row[live] = np.arange(live.size)
            assert np.array_equal(rows[2 * live.size + 1:], node_row)
            assert np.array_equal(gr.hub_node.cpu().numpy(), [0, 1, 3, 4, 5])
        xd, eed, ad, a2d = (t.to(d).requires_grad_(train) for t in (x, ee, a, a2))
        kd = keep.to(d) if drop else None
        idx = ee_index.to(d) if table else None
        if train:
            out = gat_layers.gat_heads(xd, eed, ad, a2d, gr, kd, 0.2, concat, ee_index=idx)
            (out * G.to(d)).sum().backward()
            res[frac] = [t.detach().cpu() for t in (out, xd.grad, eed.grad, ad.grad, a2d.grad)]
        else:
            with torch.no_grad():
                res[frac] = [gat_layers.gat_heads(xd, eed, ad, a2d, gr, None, 0.2, concat, ee_index=idx).cpu()]
    graph_mod.clear_graph_cache()
    names = ("out", "g_x", "g_edge_embed", "g_a", "g_a_2")
    for nm, s_, u_ in zip(names, res[0.7], res[0.0]):
        close(s_, u_, atol=2e-5, rel_to_max=2e-5, what="compacted vs all nodes " + nm)
    out, rest = res[0.7][0], res[0.7][1:]
    dead = np.setdiff1d(np.arange(N), live)
    assert not out[dead].any()
    ee_e = ee[ee_index] if table else ee
    g_x = torch.zeros_like(x); g_ee = torch.zeros_like(ee_e)
    for h in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee_e.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, concat,
                                 G[:, h * D:(h + 1) * D].double(), mask=keep[h].double() if drop else None)
        close(out[:, h * D:(h + 1) * D], r["out"].float(), what="out h%d" % h)
        if train:
            close(rest[2][h], r["g_a"].float(), atol=1e-4, what="g_a h%d" % h)
            close(rest[3][h:h + 1], r["g_a_2"].float(), atol=1e-4, what="g_a_2 h%d" % h)
            g_x += r["g_x"].float(); g_ee += r["g_edge_embed"].float()
    if train:
        close(rest[0], g_x, atol=1e-4, what="g_x")
        if table:
            g_ee = torch.zeros_like(ee).index_add_(0, ee_index, g_ee)
        close(rest[1], g_ee, atol=1e-4, what="g_edge_embed")


@pytest.mark.parametrize("N,E,live_frac", [(5000, 40000, 0.3), (5000, 40000, 0.9), (20000, 3000, 0.05), (300, 9000, 0.6), (70000, 150000, 0.2)])
def test_rows_with_edges_are_counted_and_listed(N, E, live_frac):
    """recon_graph.n_rows / row_node / rowptr_rows / node_row against numpy on both builds (the single-launch build up to 8 192 edges, the sort
    chain beyond): the rows are listed when at most ROWS_COMPACT_MAX of the nodes have in-edges, else every node stays a row."""
    from recon_amd import graph as graph_mod
    rs = np.random.RandomState(N + E)
    live = np.sort(rs.choice(N, size=max(1, int(live_frac * N)), replace=False))
    dst = live[rs.randint(0, live.size, size=E)]
    edge = torch.from_numpy(np.stack([dst, rs.randint(0, N, size=E)])).long()
    graph_mod.clear_graph_cache()
    g = graph_mod.prepare_graph(edge.to(dev()), None, N)
    have = np.unique(dst)
    if have.size <= graph_mod.ROWS_COMPACT_MAX * N:
        assert g.n_rows == have.size
        rows = g._rows.cpu().numpy()
        assert np.array_equal(rows[:have.size], have)
        deg = np.bincount(dst, minlength=N)
        assert np.array_equal(rows[have.size:2 * have.size + 1], np.concatenate([[0], np.cumsum(deg[have])]))
        node_row = np.full(N, -1); node_row[have] = np.arange(have.size)
        assert np.array_equal(rows[2 * have.size + 1:], node_row)
    else:
        assert g.n_rows == 0
    graph_mod.clear_graph_cache()


@pytest.mark.parametrize("path", ["atp", "proj"])
def test_keep_factors_taken_as_they_lie(path, monkeypatch):
    """gat_heads(keep_iid=True): H E independent dropout factors are read by the kernels in the order they lie (CSR-slot order, [E,H] for
    aggregate-then-project, [H,E] for project-then-aggregate) — equal to handing the same factors over edge by edge in original order."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    monkeypatch.setattr(gat_layers, "_GAT_PATH", path)
    d = dev()
    N, E, F_, R, D, H = 300, 4000, 24, 16, 32, 4
    g = torch.Generator().manual_seed(5)
    edge = torch.randint(0, N, (2, E), generator=g)
    x, ee = torch.randn(N, F_, generator=g), torch.randn(E, R, generator=g) * 0.5
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    K = ((torch.rand(H, E, generator=g) > 0.3).float() / 0.7).to(d)
    gr = prepare_graph(edge.to(d), None, N)
    eid = gr.eid_long
    by_edge = torch.empty(H, E, device=d)
    if path == "atp":
        by_edge[:, eid] = K.reshape(E, H).t()                            # slot k, head h reads K.reshape(E, H)[k, h]
    else:
        by_edge[:, eid] = K                                              # slot k, head h reads K[h, k]
    G = torch.randn(N, H * D, generator=g).to(d)
    res = []
    for keep, iid in ((K, True), (by_edge, False)):
        xd, eed, ad, a2d = (t.to(d).requires_grad_(True) for t in (x, ee, a, a2))
        out = gat_layers.gat_heads(xd, eed, ad, a2d, gr, keep, 0.2, True, keep_max=1.0 / 0.7, keep_iid=iid)
        (out * G).sum().backward()
        res.append([out.detach(), xd.grad, eed.grad, ad.grad, a2d.grad])
    for u, v in zip(*res):
        assert torch.equal(u, v)


@pytest.mark.parametrize("N,E,F_,R,D,H,concat,drop,nrel,extra", [
    (200, 3000, 24, 16, 32, 4, True, False, 7, 0),        # relation table only (1-hop edges), f16 x 2 capable
    (200, 3000, 24, 16, 32, 4, True, True, 7, 500),       # + 500 edges with rows of their own appended to the table (n-hop edges), dropout
    (60, 900, 10, 6, 50, 2, True, False, 3, 40),          # VEC 2
    (70, 1200, 12, 8, 20, 11, True, False, 5, 0),         # 11 heads: the per-slot gradient rows are accumulated over two head groups
    (50, 800, 200, 200, 40, 1, False, False, 9, 100),     # out_att-like
    (20, 30, 8, 8, 16, 2, True, False, 50, 0),            # more table rows than edges (the per-row score terms outgrow E x H)
])
def test_edge_embed_as_indexed_table(N, E, F_, R, D, H, concat, drop, nrel, extra, monkeypatch):
    """gat_heads(..., ee_index=...): `relation_embed[edge_type]` read in place from the table.  Outputs must be bit-equal to the call on
    the materialised E x R tensor (the kernels read the same values), input gradients equal, and the table's gradient the row sum of
    the materialised call's per-edge gradient; everything against the oracle as well.  Hub rows included."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    monkeypatch.setattr(gat_layers, "_GAT_PATH", "atp")
    d = dev()
    g = torch.Generator().manual_seed(N + E)
    dst = torch.randint(0, N, (E,), generator=g); dst[:E // 3] = 3                   # one hub destination
    src = torch.randint(0, N, (E,), generator=g); src[E // 2:E // 2 + 100] = 5       # and a hub source
    edge = torch.stack([dst, src])
    table = torch.randn(nrel + extra, R, generator=g) * 0.5
    index = torch.randint(0, nrel, (E,), generator=g)
    if extra:
        index[E - extra:] = nrel + torch.arange(extra)                                # the last `extra` edges own a row each
    x = torch.randn(N, F_, generator=g)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    keep = (torch.rand(H, E, generator=g) > 0.3).float() / 0.7 if drop else None
    G = torch.randn(N, H * D, generator=g)
    graph = prepare_graph(edge.to(d), None, N)
    assert graph.n_hub >= 1 or E < 200
    res = []
    for mode in ("table", "dense"):
        xd, td, ad, a2d = (t.to(d).requires_grad_(True) for t in (x, table, a, a2))
        kd = keep.to(d) if drop else None
        if mode == "table":
            out = gat_layers.gat_heads(xd, td, ad, a2d, graph, kd, 0.2, concat, ee_index=index.to(d))
        else:
            out = gat_layers.gat_heads(xd, gat_layers.gather_rows(td, index.to(d)), ad, a2d, graph, kd, 0.2, concat)
        (out * G.to(d)).sum().backward()
        res.append([t.detach().cpu() for t in (out, xd.grad, ad.grad, a2d.grad, td.grad)])
    assert torch.equal(res[0][0], res[1][0]), "outputs differ between table and materialised edge embeddings"
    for nm, t_, m_ in zip(("g_x", "g_a", "g_a_2", "g_table"), res[0][1:], res[1][1:]):
        close(t_, m_, atol=1e-5, rel_to_max=1e-5, what="table vs materialised " + nm)
    ee = table[index]
    g_x = torch.zeros_like(x); g_ee = torch.zeros_like(ee)
    for h in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, concat,
                                 G[:, h * D:(h + 1) * D].double(), mask=keep[h].double() if drop else None)
        close(res[0][0][:, h * D:(h + 1) * D], r["out"].float(), what="out h%d" % h)
        close(res[0][2][h], r["g_a"].float(), atol=1e-4, what="g_a h%d" % h)
        g_x += r["g_x"].float(); g_ee += r["g_edge_embed"].float()
    close(res[0][1], g_x, atol=1e-4, what="g_x")
    close(res[0][4], torch.zeros_like(table).index_add_(0, index, g_ee), atol=1e-4, what="g_table")
    with torch.no_grad():                                                            # inference call
        out_i = gat_layers.gat_heads(x.to(d), table.to(d), a.to(d), a2.to(d), graph, None, 0.2, concat, ee_index=index.to(d))
    if not drop:
        close(out_i, res[0][0], atol=1e-6, rel_to_max=1e-6, what="inference")


def test_heads_of_width_25_run_padded():
    """The reference's D = 25 per head (H * D = 200): above `_PAD_MIN_OUT` output elements `gat_heads` runs the heads 32 wide with
    zero rows appended to `a` / `a_2` and drops the extra columns; outputs and every gradient against the oracle, unpadded."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    d = dev()
    B, n, e, F_, R, D, H = 160, 16, 64, 200, 200, 25, 8
    N = B * n
    assert N * H * D >= gat_layers._PAD_MIN_OUT
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, R, seed=2)
    g = torch.Generator().manual_seed(3)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    G = torch.randn(N, H * D, generator=g)
    leaves = [t.to(d).requires_grad_(True) for t in (x, ee, a, a2)]
    out = gat_layers.gat_heads(*leaves, prepare_graph(edge.to(d), None, N), None, 0.2, True)
    assert out.shape == (N, H * D)
    (out * G.to(d)).sum().backward()
    gx, gee = torch.zeros_like(x), torch.zeros_like(ee)
    for h in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, True,
                                 G[:, h * D:(h + 1) * D].double())
        close(out[:, h * D:(h + 1) * D], r["out"].float(), what="D=25 out h%d" % h)
        close(leaves[2].grad[h], r["g_a"].float(), atol=1e-4, what="g_a h%d" % h)
        close(leaves[3].grad[h], r["g_a_2"].float().view(-1), atol=1e-4, what="g_a_2 h%d" % h)
        gx += r["g_x"].float(); gee += r["g_edge_embed"].float()
    close(leaves[0].grad, gx, atol=1e-4, what="g_x")
    close(leaves[1].grad, gee, atol=1e-4, what="g_edge_embed")


def test_cfg5_bf16_gat_then_gcn_stack():
    """BASELINE.json configs[4] in its stated dtype: power-law graphs, an H-head GAT layer with bf16 features in and out (fp32
    kernels inside), followed by a 3-layer bf16 GraphConvolution stack on the same node features, forward and backward.  Against
    the fp32 oracle on the SAME bf16-rounded inputs: what remains is the rounding of each layer's result to bf16 (2^-8 relative)."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    from recon_amd.gcn_layers import GraphConvolution
    d = dev()
    edge, N = _power_law_batch(4, seed=3, max_n=128)
    E = edge.shape[1]
    F_, R, D, H = 32, 16, 16, 4
    g = torch.Generator().manual_seed(0)
    bf = lambda t: t.to(torch.bfloat16)
    x = bf(torch.randn(N, F_, generator=g))
    ee = bf(torch.randn(E, R, generator=g) * 0.5)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    xd, eed = x.to(d).requires_grad_(True), ee.to(d).requires_grad_(True)
    ad, a2d = a.to(d).requires_grad_(True), a2.to(d).requires_grad_(True)
    h = gat_layers.gat_heads(xd, eed, ad, a2d, prepare_graph(edge.to(d), None, N), None, 0.2, True)
    assert h.dtype == torch.bfloat16 and h.shape == (N, H * D)
    refs = [O.gat_layer_forward(x.double(), edge, ee.double(), None, None, a[i].double(), a2[i:i + 1].double(), 0.2, True) for i in range(H)]
    ref_h = torch.cat(refs, dim=1).float()
    close(h.float(), ref_h, atol=2e-3, rel_to_max=1e-2, what="bf16 GAT heads")
    # the same node features through three bf16 graph convolutions over a dense row-normalised adjacency of the batch
    torch.manual_seed(2)
    adj = torch.zeros(N, N)
    adj[edge[0], edge[1]] = 1.0
    adj += torch.eye(N)
    adj = bf(adj / adj.sum(-1, keepdim=True))
    layers = [GraphConvolution(H * D, H * D).to(torch.bfloat16).to(d) for _ in range(3)]
    cur, ref = h, bf(ref_h).float()
    for l in layers:
        cur = l(cur, adj.to(d))
        ref = bf(O.graph_convolution(ref, adj.float(), l.weight.detach().float().cpu(), l.bias.detach().float().cpu())).float()
    assert cur.dtype == torch.bfloat16
    close(cur.float(), ref, atol=5e-3, rel_to_max=5e-2, what="bf16 GAT + 3 x GCN")
    cur.float().sum().backward()                                  # gradients reach every input of the mixed stack, in its dtype
    assert xd.grad.dtype == torch.bfloat16 and eed.grad.dtype == torch.bfloat16 and ad.grad.dtype == torch.float32
    for t in (xd.grad, eed.grad, ad.grad, a2d.grad, layers[0].weight.grad):
        assert bool(torch.isfinite(t.float()).all()) and float(t.float().abs().max()) > 0


def test_cfg5_bf16_attention_at_the_benched_shapes():
    """The attention leg of the benched configs[4] stack (tools/secondary.py powerlaw_mixed_stack_bf16: F = R = 200, D = 32, H = 8, 64
    power-law graphs of up to 256 nodes) with bfloat16 x / edge_embed read IN PLACE by the forward kernels (recon_gat_atp_args.io_bf16)
    against the float64 oracle on the same bf16-rounded inputs: every head's forward; the input gradients for an upstream gradient that
    lives in heads 0 and 7 (the oracle's two heads' contributions); parameters' gradients of those heads at the float32 bar."""
    from recon_amd import gat_layers, _lib
    from recon_amd.graph import prepare_graph
    d = dev()
    edge, N = _power_law_batch(64, seed=0, max_n=256)
    E = edge.shape[1]
    F_, R, D, H = 200, 200, 32, 8
    assert _lib.lib().recon_gat_atp_bf16_io_supported(F_, R, D, H) == 1
    g = torch.Generator().manual_seed(1)
    bf = lambda t: t.to(torch.bfloat16)
    x, ee = bf(torch.randn(N, F_, generator=g)), bf(torch.randn(E, R, generator=g) * 0.5)
    a = torch.randn(H, D, 3 * F_, generator=g) * (2.0 / (3 * F_ + D)) ** 0.5
    a2 = torch.randn(H, D, generator=g) * (2.0 / (D + 1)) ** 0.5
    xd, eed = x.to(d).requires_grad_(True), ee.to(d).requires_grad_(True)
    ad, a2d = a.to(d).requires_grad_(True), a2.to(d).requires_grad_(True)
    graph = prepare_graph(edge.to(d), None, N)
    assert graph.n_hub > 0                                            # power-law rows: the hub pieces' kernels read bf16 rows too
    h = gat_layers.gat_heads(xd, eed, ad, a2d, graph, None, 0.2, True)
    assert h.dtype == torch.bfloat16 and h.shape == (N, H * D)
    G = torch.zeros(N, H * D)
    G[:, :D] = torch.randn(N, D, generator=g)
    G[:, 7 * D:] = torch.randn(N, D, generator=g)
    h.backward(bf(G).to(d))
    g_x, g_ee = 0, 0
    for i in range(H):
        if i in (0, 7):
            r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[i].double(), a2[i:i + 1].double(), 0.2, True, bf(G[:, i * D:(i + 1) * D]).double())
            ref_i = r["out"]
            g_x, g_ee = g_x + r["g_x"], g_ee + r["g_edge_embed"]
            close(ad.grad[i], r["g_a"].float(), atol=1e-5, rel_to_max=1e-4, what="benched cfg5 g_a head %d" % i)
            close(a2d.grad[i], r["g_a_2"].float().reshape(-1), atol=1e-5, rel_to_max=1e-4, what="benched cfg5 g_a_2 head %d" % i)
        else:
            ref_i = O.gat_layer_forward(x.double(), edge, ee.double(), None, None, a[i].double(), a2[i:i + 1].double(), 0.2, True)
        close(h.float()[:, i * D:(i + 1) * D], ref_i.float(), atol=1e-5, rel_to_max=2.0 ** -8, what="benched cfg5 out head %d (one rounding to bf16)" % i)
    assert xd.grad.dtype == torch.bfloat16 and eed.grad.dtype == torch.bfloat16
    close(xd.grad.float(), g_x.float(), atol=1e-5, rel_to_max=2.0 ** -8, what="benched cfg5 g_x (bf16 out)")
    close(eed.grad.float(), g_ee.float(), atol=1e-5, rel_to_max=2.0 ** -8, what="benched cfg5 g_edge_embed (bf16 out)")


@pytest.mark.parametrize("name", ["spkbgat1_nhop", "spkbgat2_1hop"])
def test_spkbgat_golden(name):
    """G7: the stage-A model (whole entity table, one entity batch of edges) vs the reference SpKBGATModified:
    strict state_dict load, forward, in-place side effects, all gradients, batch_test."""
    from recon_amd.models import SpKBGATModified
    g = load_golden(name)
    d = dev()
    H, nhid = int(g["nheads"]), int(g["nhid"])
    sd0 = {k[3:]: T(g[k]) for k in g if k.startswith("p0.")}
    m = SpKBGATModified(sd0["entity_embeddings"].clone(), sd0["relation_embeddings"].clone(), [nhid, nhid * H],
                        [nhid * H, nhid * H], 0.0, float(g["alpha"]), [H, H], None)
    m.load_state_dict(sd0, strict=True)
    m = m.to(d).eval()
    adj = (T(g["edge"]).to(d), T(g["edge_type"]).to(d))
    nhop = T(g["nhop"]).to(d)
    be = T(g["batch_entities"]).to(d)
    out_e, out_r, mask = m(None, be, adj, nhop)
    close(out_e, g["out_entity"], what="out_entity")
    close(out_r, g["out_relation"], what="out_relation")
    np.testing.assert_array_equal(mask.cpu().numpy(), g["mask"])
    ((out_e * T(g["G"]).to(d)).sum() + (out_r * T(g["G2"]).to(d)).sum()).backward()
    for k, p in m.named_parameters():
        if "g." + k in g:
            close(p.grad, g["g." + k], atol=1e-5, what="g." + k)
    sd1 = m.state_dict()
    for k in ("entity_embeddings", "final_entity_embeddings", "final_relation_embeddings"):
        close(sd1[k], g["p1." + k], atol=2e-5, what="side effect " + k)
    with torch.no_grad():
        te, tr, _ = m.batch_test(None, be, adj, nhop, T(g["test_input"]).to(d))
    close(te, g["test_entity"], what="batch_test entity")
    close(tr, g["test_relation"], what="batch_test relation")


@pytest.mark.parametrize("prune", [True, False])
@pytest.mark.parametrize("family", ["2", "1", "0"])
@pytest.mark.parametrize("path", ["atp", "proj"])
def test_spkbgat_train_mode_three_sgd_iterations_golden(family, path, prune, monkeypatch):
    """The regime stage A runs (GAT/main.py:478-525): SpKBGATModified with drop_GAT = 0.3 in train(), three iterations of
    forward -> batch_gat_loss -> backward -> SGD(lr = 1e-3) on three different batches, against the REFERENCE's losses and final parameters.
    The reference's dropout factors were recorded in call order — one E-vector per head (GAT/layers.py:158 inside GAT/models.py:71-72),
    dropout_layer on the concatenated heads (:73), out_att's E-vector (:86) — and are replayed here through draw_keep / dropout_layer.
    Every GEMM family (f16 x 2, bf16 x 3, exact fp32) and both formulations; with and without the pruning of the edges into rows the
    model's mask discards (models.PRUNE_DEAD_ROWS: the surviving edges take their factors from the recorded vectors by position)."""
    from recon_amd import gat_layers, models
    from recon_amd.models import SpKBGATModified
    from recon_amd.losses import batch_gat_loss
    monkeypatch.setattr(gat_layers, "_GEMM_BX3", family)
    monkeypatch.setattr(gat_layers, "_GAT_PATH", path)
    monkeypatch.setattr(models, "PRUNE_DEAD_ROWS", prune)
    monkeypatch.setattr(models, "KEEP_PRUNED_POSITIONS", True)
    g = load_golden("spkbgat3_train")
    d = dev()
    H, nhid, ratio = int(g["nheads"]), int(g["nhid"]), int(g["ratio"])
    sd0 = {k[3:]: T(g[k]) for k in g if k.startswith("p0.")}
    m = SpKBGATModified(sd0["entity_embeddings"].clone(), sd0["relation_embeddings"].clone(), [nhid, nhid * H], [nhid * H, nhid * H],
                        float(g["p_drop"]), float(g["alpha"]), [H, H], None)
    m.load_state_dict(sd0, strict=True)
    m = m.to(d).train()
    sg = m.sparse_gat_1
    assert abs(sg.attentions[0].keep_bound() - 1.0 / (1.0 - float(g["p_drop"]))) < 1e-6
    opt = torch.optim.SGD(m.parameters(), lr=float(g["lr"]))
    loss_fn = torch.nn.MarginRankingLoss(margin=float(g["margin"]))
    drawn = []
    kept = lambda mk: mk if m._pruned_pos is None else mk.reshape(-1)[m._pruned_pos]       # the factors of the edges the model kept
    n_kept = []
    for it in range(3):
        masks = [T(g["it%d.mask%d" % (it, k)]).to(d) for k in range(H + 2)]
        assert float(masks[0].max()) <= sg.attentions[0].keep_bound() + 1e-6
        for h, att in enumerate(sg.attentions):
            att.draw_keep = (lambda mk, tag: (lambda E, device: (drawn.append(tag), kept(mk).view(1, E))[1]))(masks[h], "head%d" % h)
        sg.dropout_layer.forward = (lambda mk: (lambda x: (drawn.append("layer"), x * mk)[1]))(masks[H])
        sg.out_att.draw_keep = (lambda mk: (lambda E, device: (drawn.append("out"), kept(mk).view(1, E))[1]))(masks[H + 1])
        drawn.clear()
        out_e, out_r, _ = m(None, T(g["it%d.batch_entities" % it]).to(d), (T(g["it%d.edge" % it]).to(d), T(g["it%d.edge_type" % it]).to(d)),
                            T(g["it%d.nhop" % it]).to(d))
        assert drawn == ["head%d" % h for h in range(H)] + ["layer", "out"], drawn          # the reference's draw order
        assert (m._pruned_pos is not None) == prune
        n_kept.append(m._pruned_pos.numel() if prune else masks[0].numel())
        close(out_e, g["it%d.out_entity" % it], atol=2e-5, what="train-mode out_entity, iteration %d" % it)
        close(out_r, g["it%d.out_relation" % it], atol=2e-5, what="train-mode out_relation, iteration %d" % it)
        opt.zero_grad()
        loss = batch_gat_loss(loss_fn, T(g["it%d.train_indices" % it]).to(d), out_e, out_r, valid_invalid_ratio_gat=ratio)
        loss.backward()
        opt.step()
        np.testing.assert_allclose(loss.item(), g["losses"][it], rtol=1e-4, err_msg="loss of iteration %d" % it)
    if prune:
        assert sum(n_kept) < sum(int(g["it%d.mask0" % it].size) for it in range(3))      # the fixture's batches do have rows the mask discards
    for k, v in m.state_dict().items():
        np.testing.assert_allclose(v.cpu().numpy(), g["p3." + k], atol=1e-5, rtol=0, err_msg=k)
        if k not in ("entity_embeddings", "final_entity_embeddings", "final_relation_embeddings"):
            d_ref = g["p3." + k] - g["p0." + k]                                               # the three SGD updates themselves
            np.testing.assert_allclose(v.cpu().numpy() - g["p0." + k], d_ref, atol=5e-3 * np.abs(d_ref).max() + 1e-9, err_msg="delta " + k)


def test_sep_space_spkbgat_golden():
    """GAT_sep_space: the stage-A model with W_ent2rel — strict state_dict load, forward, the relation-space projection of every
    triple's entities (GAT_sep_space/main.py:359-367) grouped by relation instead of gathered per triple, and every gradient of the
    L1 translation residual (W_ent2rel's included) against the reference."""
    from recon_amd.sep_space import SpKBGATModified
    g = load_golden("sepspace1")
    d = dev()
    H, nhid = int(g["nheads"]), int(g["nhid"])
    sd0 = {k[3:]: T(g[k]) for k in g if k.startswith("p0.")}
    m = SpKBGATModified(sd0["entity_embeddings"].clone(), sd0["relation_embeddings"].clone(), [nhid, nhid * H], [nhid * H, nhid * H], 0.0, 0.2, [H, H], None)
    m.load_state_dict(sd0, strict=True)
    m = m.to(d).eval()
    out_e, out_r, mask = m(None, T(g["batch_entities"]).to(d), (T(g["edge"]).to(d), T(g["edge_type"]).to(d)), T(g["nhop"]).to(d))
    close(out_e, g["out_entity"], what="out_entity")
    close(out_r, g["out_relation"], what="out_relation")
    tri = T(g["triples"]).to(d)
    src = m.ent2rel(out_e[tri[:, 0]], tri[:, 1])
    dst = m.ent2rel(out_e[tri[:, 2]], tri[:, 1])
    close(src, g["src_rel"], atol=2e-5, what="tanh(e_src W_r)")
    close(dst, g["dst_rel"], atol=2e-5, what="tanh(e_dst W_r)")
    norm = torch.norm(src + out_r[tri[:, 1]] - dst, p=1, dim=1)
    close(norm, g["norm"], atol=1e-4, what="L1 residual")
    (norm * T(g["Gn"]).to(d)).sum().backward()
    seen = 0
    for k, p in m.named_parameters():
        if "g." + k in g:
            close(p.grad, g["g." + k], atol=2e-5, rel_to_max=1e-4, what="g." + k)
            seen += 1
    assert seen == sum(k.startswith("g.") for k in g) and "g.W_ent2rel" in g
    assert m.ent2rel(out_e[:0], tri[:0, 1]).shape == (0, nhid * H)


@pytest.mark.parametrize("N,E,C_,skew", [(5, 4000, 24, True), (300, 1000, 7, False), (64, 70000, 200, True), (1000, 3, 4, False), (237, 50000, 50, True), (9, 0, 5, False),
                                          (40, 3000, 130, False)])          # C = 50, 130: the two-wide column form
def test_spmm_rowsum_long_and_short_segments(N, E, C_, skew):
    """SpecialSpmmFinal on segments from empty to tens of thousands of edges (a relation type owning most edges)."""
    from recon_amd.gat_layers import SpecialSpmmFinal
    g = torch.Generator().manual_seed(E + N)
    dst = torch.randint(0, N, (E,), generator=g)
    if skew:
        dst[: E * 3 // 4] = 1 % N                                   # one destination owns 3/4 of all edges
        dst = dst[torch.randperm(E, generator=g)]
    edge = torch.stack([dst, torch.randint(0, N, (E,), generator=g)])
    w = torch.randn(E, C_, generator=g)
    out = SpecialSpmmFinal()(edge.to(dev()), w.to(dev()), N, E, C_)
    ref = O.spmm_rowsum(edge, w.double(), N)
    close(out, ref.float(), atol=1e-5, rel_to_max=2e-6, what="rowsum")


def test_spmm_rowsum_beyond_2_31_elements():
    """SpecialSpmmFinal with E x out_features = 2.4 G elements of edge values (9.8 GB): row offsets need 64-bit arithmetic.  Two copies
    of one edge list with the same values: both halves of the output must agree, and a sample of rows with float64 sums."""
    from recon_amd.gat_layers import SpecialSpmmFinal
    d = dev()
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~40 GB of device memory")
    Nh, Eh, C_ = 300000, 2400000, 512
    assert 2 * Eh * C_ > 2 ** 31
    g = torch.Generator().manual_seed(3)
    dst = torch.randint(0, Nh, (Eh,), generator=g)
    dst[:5000] = 11
    w = torch.randn(Eh, C_, generator=g)
    edge = torch.stack([torch.cat([dst, dst + Nh]), torch.zeros(2 * Eh, dtype=torch.long)])
    wd = torch.cat([w, w]).to(d).requires_grad_(True)
    out = SpecialSpmmFinal()(edge.to(d), wd, 2 * Nh, 2 * Eh, C_)
    assert torch.equal(out[:Nh], out[Nh:])
    rows = torch.cat([torch.tensor([11]), torch.randint(0, Nh, (200,), generator=g)]).unique()
    ref = torch.stack([w[dst == r].double().sum(0) for r in rows])
    close(out[rows.to(d)], ref.float(), atol=1e-4, rel_to_max=2e-6, what="sampled rows")
    G = torch.randn(2 * Nh, C_, generator=g).to(d)
    out.backward(G)
    pick = torch.randint(0, 2 * Eh, (1000,), generator=g)
    assert torch.equal(wd.grad[pick.to(d)], G[edge[0][pick].to(d)])      # GAT/layers.py:67-79: grad_edge_w[e] = grad_out[edge[0, e]]


@pytest.mark.parametrize("N,E,F_,R,D,H", [
    (16, 0, 8, 8, 16, 2),             # no edges at all
    (40, 7, 200, 200, 200, 8),        # nearly empty graph at cfg-2 widths (most rows isolated: Z clamp path)
    (300, 1200, 200, 200, 200, 8),    # cfg-2 widths, ragged sizes (row / column tails of every GEMM tile)
    (33, 100, 12, 20, 24, 3),         # odd head count, D % 8 == 0 but tiny
    (64, 256, 16, 16, 8, 8),          # D = 8: one 16-byte plane slot per head
    (129, 500, 40, 36, 208, 2),       # D = 208: exactly one full column tile
    (500, 100, 200, 200, 200, 8),     # N >> E: the project-then-aggregate kernels take over
])
def test_training_edge_shapes(N, E, F_, R, D, H):
    """Forward + all four gradients of the fused H-head stage against the oracle on shapes that hit the tile tails, the
    empty-input paths and the formulation switch."""
    from recon_amd.gat_layers import gat_heads
    from recon_amd.graph import prepare_graph
    g = torch.Generator().manual_seed(N + E)
    x, ee = torch.randn(N, F_, generator=g), torch.randn(E, R, generator=g)
    edge = torch.randint(0, N, (2, E), generator=g)
    a, a2 = torch.randn(H, D, 2 * F_ + R, generator=g) * 0.1, torch.randn(H, D, generator=g) * 0.1
    G = torch.randn(N, H * D, generator=g)
    leaves = [t.to(dev()).requires_grad_(True) for t in (x, ee, a, a2)]
    out = gat_heads(*leaves, prepare_graph(edge.to(dev()), None, N), None, 0.2, True)
    grads = torch.autograd.grad(out, leaves, G.to(dev()))
    cl = [t.clone().requires_grad_(True) for t in (x, ee, a, a2)]
    ref = torch.cat([O.gat_layer_forward(cl[0], edge, cl[1], None, None, cl[2][h], cl[3][h:h + 1], 0.2, True) for h in range(H)], 1)
    rg = torch.autograd.grad(ref, cl, G)
    close(out, ref, what="out")
    for name, u, v in zip(("g_x", "g_edge_embed", "g_a", "g_a_2"), grads, rg):
        if v.numel():
            close(u, v, atol=1e-4, rel_to_max=1e-4, what=name)


def test_phased_backward_on_two_streams_matches(monkeypatch):
    """recon_gat_atp_bwd_phase (PREPARE -> {INPUTS | WEIGHTS on a side stream} -> FINISH) produces exactly what the
    single-call backward does."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    N, E, F_, R, D, H = 300, 1200, 40, 24, 48, 4
    g = torch.Generator().manual_seed(3)
    x, ee = torch.randn(N, F_, generator=g), torch.randn(E, R, generator=g)
    edge = torch.randint(0, N, (2, E), generator=g)
    a, a2 = torch.randn(H, D, 2 * F_ + R, generator=g) * 0.1, torch.randn(H, D, generator=g) * 0.1
    G = torch.randn(N, H * D, generator=g).to(dev())
    graph = prepare_graph(edge.to(dev()), None, N)
    res = []
    for overlap in (False, True):
        monkeypatch.setattr(gat_layers, "_OVERLAP", overlap)
        leaves = [t.to(dev()).requires_grad_(True) for t in (x, ee, a, a2)]
        out = gat_layers.gat_heads(*leaves, graph, None, 0.2, True)
        res.append([t.detach().clone() for t in torch.autograd.grad(out, leaves, G)])
        torch.cuda.synchronize()
    for u, v in zip(*res):
        assert torch.equal(u, v)



def test_spgat_forward_with_bf16_features_and_relation_table():
    """SpGAT.forward with bf16 entity features and a bf16 relation table (reduced-precision storage at the layer boundary): the models hand
    the layers `IndexedRows(table, edge_type)`; with a non-fp32 table the rows are materialised by index_select.  Against the fp32 oracle on
    the bf16-rounded inputs (each layer's result is rounded to bf16 once)."""
    from recon_amd.models import SpGAT
    d = dev()
    N, E, F_, D, H, nrel = 96, 400, 32, 16, 2, 7
    g = torch.Generator().manual_seed(4)
    x = torch.randn(N, F_, generator=g).to(torch.bfloat16)
    rel = torch.randn(nrel, F_, generator=g).to(torch.bfloat16)
    edge = torch.randint(0, N, (2, E), generator=g)
    et = torch.randint(0, nrel, (E,), generator=g)
    torch.manual_seed(0)
    m = SpGAT(N, F_, D, F_, 0.0, 0.2, H).to(d)
    out, out_rel = m(None, x.to(d), rel.to(d), edge.to(d), et.to(d), None, torch.tensor([]), torch.tensor([]))
    assert out.shape == (N, H * D) and torch.isfinite(out.float()).all()
    sd = {k: v.detach().cpu().float() for k, v in m.state_dict().items()}
    ref, ref_rel = O.spgat_forward(x.float(), rel.float(), edge, et, rel.float()[et], None, None,
                                   [sd["attention_%d.a" % h] for h in range(H)], [sd["attention_%d.a_2" % h] for h in range(H)],
                                   sd["W"], sd["out_att.a"], sd["out_att.a_2"], 0.2)
    close(out.float(), ref, atol=2e-2, rel_to_max=3e-2, what="bf16 SpGAT out")


def test_weight_gradient_early_sum_schedule_matches(gemm_family):
    """RECON_ATP_BWD_EARLY_SUM (the data-parallel backward: G = V^T g_h summed over split-K and handed to the collective BEFORE the
    edge chain, g_a += a_2 (x) g_u afterwards) against the single-pass backward, with a stand-in reducer that records the order of
    the calls and leaves the tensors alone (one rank's mean is itself).  Same gradients to fp32 round-off (fmaf vs mul + add in
    the last kernel), input gradients bit-equal."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    N, E, F_, R, D, H = 2048, 8192, 64, 64, 64, 4
    g = torch.Generator().manual_seed(5)
    x, ee = torch.randn(N, F_, generator=g), torch.randn(E, R, generator=g)
    edge = torch.randint(0, N, (2, E), generator=g)
    a, a2 = torch.randn(H, D, 2 * F_ + R, generator=g) * 0.1, torch.randn(H, D, generator=g) * 0.1
    G = torch.randn(N, H * D, generator=g).to(dev())
    graph = prepare_graph(edge.to(dev()), None, N)

    class Recorder:
        def __init__(self):
            self.calls, self.reduced = [], []

        def all_reduce_mean(self, t, async_op=False):
            self.calls.append((tuple(t.shape), async_op))
            return None

        def mark_reduced(self, *ts):
            self.reduced.extend(ts)
    res = []
    for sync in (None, Recorder()):
        prev = gat_layers.set_weight_grad_sync(sync)
        try:
            leaves = [t.to(dev()).requires_grad_(True) for t in (x, ee, a, a2)]
            out = gat_layers.gat_heads(*leaves, graph, None, 0.2, True)
            res.append([t.detach().clone() for t in torch.autograd.grad(out, leaves, G)])
        finally:
            gat_layers.set_weight_grad_sync(prev)
    assert sync.calls == [((H, D, 2 * F_ + R), True), ((H * (2 * F_ + R),), False)] and len(sync.reduced) == 2
    assert torch.equal(res[0][0], res[1][0]) and torch.equal(res[0][1], res[1][1])
    close(res[1][2], res[0][2], atol=1e-6, rel_to_max=1e-6, what="g_a, early-sum schedule")
    close(res[1][3], res[0][3], atol=1e-6, rel_to_max=1e-6, what="g_a_2, early-sum schedule")


@pytest.mark.parametrize("xs,es,gs,p", [(1e3, 1e-3, 1e-8, 0.0), (1e-4, 1e2, 1e6, 0.0), (1.0, 1.0, 1.0, 0.9), (30.0, 1e-2, 1e-3, 0.5)])
def test_wide_dynamic_range_inputs(xs, es, gs, p):
    """The f16 x 2 GEMM family lives on per-tensor power-of-two scales (half has 5 exponent bits): inputs, gradients and
    dropout factors far from unit scale must come out as accurate, relative to each tensor's own magnitude, as at unit scale."""
    from recon_amd.gat_layers import gat_heads
    from recon_amd.graph import prepare_graph
    N, E, F_, R, D, H = 96, 400, 24, 16, 32, 4
    g = torch.Generator().manual_seed(17)
    x, ee = torch.randn(N, F_, generator=g) * xs, torch.randn(E, R, generator=g) * es
    edge = torch.randint(0, N, (2, E), generator=g)
    a = torch.randn(H, D, 2 * F_ + R, generator=g) * (0.05 / max(xs, es))       # keeps the scores O(1)
    a2 = torch.randn(H, D, generator=g)
    G = torch.randn(N, H * D, generator=g) * gs
    keep = ((torch.rand(H, E, generator=g) >= p).float() / (1.0 - p)) if p > 0 else None
    leaves = [t.to(dev()).requires_grad_(True) for t in (x, ee, a, a2)]
    out = gat_heads(*leaves, prepare_graph(edge.to(dev()), None, N), keep.to(dev()) if keep is not None else None, 0.2, True,
                    keep_max=(1.0 / (1.0 - p)) if p > 0 else None)
    grads = torch.autograd.grad(out, leaves, G.to(dev()))
    cl = [t.double().clone().requires_grad_(True) for t in (x, ee, a, a2)]
    ref = torch.cat([O.gat_layer_forward(cl[0], edge, cl[1], None, None, cl[2][h], cl[3][h:h + 1], 0.2, True,
                                         mask=keep[h].double() if keep is not None else None) for h in range(H)], 1)
    rg = torch.autograd.grad(ref, cl, G.double())
    assert torch.isfinite(out).all()
    close(out, ref.float(), atol=0.0, rel_to_max=2e-5, what="out")
    for name, u, v in zip(("g_x", "g_edge_embed", "g_a", "g_a_2"), grads, rg):
        assert torch.isfinite(u).all(), name
        close(u, v.float(), atol=0.0, rel_to_max=1e-4, what=name)


@pytest.mark.gpu
@pytest.mark.parametrize("T,D,Dout,R,skew", [
    (1000, 200, 200, 237, "zipf"),      # GAT_sep_space's shape: entity_out_dim_1 * nheads = 200, FB15k-237's relations, few frequent ones
    (37, 48, 24, 5, "uniform"),         # T % 16 != 0, rectangular matrices
    (300, 64, 300, 300, "uniform"),     # about one row per relation: every tile holds many runs; Dout > 256 (two columns per thread)
    (64, 8, 8, 3, "single"),            # one relation for all rows, two relations without rows
    (0, 16, 16, 4, "uniform"),          # no rows
])
def test_rel_rows_mm_vs_bmm(T, D, Dout, R, skew):
    """recon_rel_rows_mm / _wgrad (csrc/rel_mm.hip) against the reference's own expression `torch.bmm(x.unsqueeze(1), W[rel])`
    (GAT_sep_space/main.py:359-364): forward, d x, d W (relations without rows: zeros)."""
    from recon_amd.sep_space import rel_rows_mm
    d = dev()
    g = torch.Generator().manual_seed(T + D)
    x = torch.randn(T, D, generator=g)
    W = torch.randn(R, D, Dout, generator=g) / D ** 0.5
    if skew == "zipf":
        p = 1.0 / torch.arange(1, R + 1, dtype=torch.float64)
        rel = torch.multinomial(p / p.sum(), T, replacement=True, generator=g)
    elif skew == "single":
        rel = torch.full((T,), 1, dtype=torch.int64)
    else:
        rel = torch.randint(0, R, (T,), generator=g)
    G = torch.randn(T, Dout, generator=g)
    xr, Wr = x.double().requires_grad_(True), W.double().requires_grad_(True)
    ref = torch.bmm(xr.unsqueeze(1), Wr[rel]).squeeze(1)
    (ref * G.double()).sum().backward()
    xd, Wd = x.to(d).requires_grad_(True), W.to(d).requires_grad_(True)
    out = rel_rows_mm(xd, rel.to(d), Wd)
    (out * G.to(d)).sum().backward()
    close(out, ref.detach().float(), atol=1e-5, rel_to_max=1e-5, what="rel_rows_mm out")
    close(xd.grad, xr.grad.float(), atol=1e-5, rel_to_max=1e-5, what="rel_rows_mm g_x")
    close(Wd.grad, Wr.grad.float(), atol=1e-5, rel_to_max=1e-5, what="rel_rows_mm g_W")
    out2 = rel_rows_mm(xd.detach(), rel.to(d), Wd.detach())
    assert torch.equal(out2, out.detach()), "not deterministic"


@pytest.mark.gpu
def test_head_gradients_written_into_the_bucket():
    """models.SpGAT.write_head_gradients_into(bucket): the heads' backward writes d a / d a_2 into the flat gradient buffer itself — the
    parameters' gradients are views of it, pack() copies nothing, the values equal the plain path's bit for bit; a second backward without
    zero() (gradient accumulation) must still add, not alias."""
    from recon_amd.models import SpGAT
    from recon_amd.dist import FlatGradBucket
    from recon_amd import synth
    d = dev()
    B, n, e, F_, D, H = 6, 8, 24, 16, 16, 4
    N = B * n
    x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=3)
    nohop = torch.tensor([])
    G = torch.randn(N, H * D, generator=torch.Generator().manual_seed(5)).to(d)

    def make():
        torch.manual_seed(0)
        return SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(d)
    plain, bound = make(), make()
    pb = FlatGradBucket([p for att in plain.attentions for p in (att.a, att.a_2)])
    bb = FlatGradBucket(bound.head_parameters())
    assert bound.write_head_gradients_into(bb)
    assert not make().write_head_gradients_into(FlatGradBucket([p for att in plain.attentions for p in (att.a, att.a_2)]))   # foreign / interleaved layout
    xd, eed, ed = x.to(d), ee.to(d), edge.to(d)
    for model, bucket in ((plain, pb), (bound, bb)):
        bucket.zero()
        model.heads_forward(xd, ed, eed, nohop, nohop).backward(G)
    lo, hi = bb.flat.data_ptr(), bb.flat.data_ptr() + 4 * bb.flat.numel()
    assert all(lo <= p.grad.data_ptr() < hi for p in bound.head_parameters()), "gradients are not views of the bucket"
    before = bb.flat.clone()
    bb.allreduce_mean()
    assert torch.equal(bb.flat, before)                                 # nothing to copy, world size 1
    pb.allreduce_mean()
    for a, b in zip(plain.attentions, bound.attentions):
        assert torch.equal(a.a.grad, b.a.grad) and torch.equal(a.a_2.grad, b.a_2.grad)
    # accumulation: a second backward while the gradients are still set
    first = [p.grad.clone() for p in bound.head_parameters()]
    bound.heads_forward(xd, ed, eed, nohop, nohop).backward(G)
    for p, g1 in zip(bound.head_parameters(), first):
        close(p.grad, 2 * g1, atol=1e-6, rel_to_max=1e-6, what="accumulated head gradient")
    # two forwards before one backward (loss(batch 1) + loss(batch 2), positive / negative passes, a checkpoint re-forward): both forwards see
    # .grad None, both backwards run inside ONE autograd pass — only the first may take the bucket region, the other must add to it
    x2 = (xd * 0.5 + 0.25).contiguous()
    grads = {}
    for name, model, bucket in (("plain", plain, pb), ("bound", bound, bb)):
        bucket.zero()
        o1 = model.heads_forward(xd, ed, eed, nohop, nohop)
        o2 = model.heads_forward(x2, ed, eed, nohop, nohop)
        ((o1 * G).sum() + (o2 * G).sum()).backward()
        grads[name] = [p.grad.clone() for p in (model.head_parameters() if name == "bound" else [p for att in model.attentions for p in (att.a, att.a_2)])]
    ref = {id(p): g for p, g in zip([p for att in plain.attentions for p in (att.a, att.a_2)], grads["plain"])}
    for att_p, att_b in zip(plain.attentions, bound.attentions):
        for pp, pbnd in ((att_p.a, att_b.a), (att_p.a_2, att_b.a_2)):
            close(pbnd.grad, ref[id(pp)], atol=1e-6, rel_to_max=1e-6, what="head gradient of two forwards under one backward")


@pytest.mark.gpu
def test_device_nan_flag_is_raised_without_a_round_trip():
    """The reference asserts `not torch.isnan(...)` three times per layer call (GAT/layers.py:147, :167, :172: three host syncs); here the
    kernels raise a device word instead and nobody waits for it.  Clean inputs leave it zero on both formulations; a NaN feature row, an
    overflowing score (exp(-leakyrelu) = inf: the reference's h_prime = inf / inf) and a NaN weight raise it."""
    from recon_amd import gat_layers
    from recon_amd.gat_layers import gat_heads, nan_raised, enable_nan_flag
    from recon_amd.graph import prepare_graph
    d = dev()
    enable_nan_flag(d)
    B, n, e, F_, D, H = 4, 8, 24, 16, 16, 2
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, F_, seed=1)
    g = torch.Generator().manual_seed(2)
    a = torch.randn(H, D, 3 * F_, generator=g) * 0.1
    a2 = torch.randn(H, D, generator=g) * 0.1
    graph = prepare_graph(edge.to(d), None, B * n)
    for path in ("atp", "proj"):
        saved = gat_layers._GAT_PATH
        gat_layers._GAT_PATH = path
        try:
            assert not nan_raised(d)
            out = gat_heads(x.to(d), ee.to(d), a.to(d), a2.to(d), graph, None, 0.2, True)
            assert torch.isfinite(out).all() and not nan_raised(d), path
            xb = x.clone(); xb[3, 5] = float("nan")
            gat_heads(xb.to(d), ee.to(d), a.to(d), a2.to(d), graph, None, 0.2, True)
            assert nan_raised(d), path + ": NaN feature"
            assert not nan_raised(d), "the word is cleared by the read"
            gat_heads(x.to(d), ee.to(d), (a * 1e4).to(d), (a2 * 1e4).to(d), graph, None, 0.2, True)      # scores of -1e6: exp overflows
            assert nan_raised(d), path + ": overflowing weights"
            ab = a.clone(); ab[1, 2, 3] = float("nan")
            gat_heads(x.to(d), ee.to(d), ab.to(d), a2.to(d), graph, None, 0.2, True)
            assert nan_raised(d), path + ": NaN parameter"
        finally:
            gat_layers._GAT_PATH = saved


@pytest.mark.gpu
@pytest.mark.parametrize("rows,width,E", [(237, 50, 25849), (237, 100, 4001), (64, 37, 513), (5, 3, 7), (9, 200, 0)])
def test_gather_rows_pair_is_the_sum_of_two_gathers(rows, width, E):
    """relation_embed[t[:, 0]] + relation_embed[t[:, 1]] (GAT/models.py:64-65, 80-81): forward bit-equal to the two indexings added, the
    table gradient equal to index_add over both columns (float64 reference); widths for each vector form, int32 indices, no rows."""
    from recon_amd import gat_layers
    d = torch.device("cuda:0")
    gen = torch.Generator().manual_seed(rows + width)
    table = torch.randn(rows, width, generator=gen).to(d).requires_grad_(True)
    idx = torch.randint(0, rows, (E, 2), generator=gen).to(d)
    out = gat_layers.gather_rows_pair(table, idx)
    assert out.shape == (E, width)
    assert torch.equal(out, table.detach()[idx[:, 0]] + table.detach()[idx[:, 1]])
    assert torch.equal(gat_layers.gather_rows_pair(table.detach(), idx.int()), out)
    G = torch.randn(E, width, generator=gen).to(d)
    out.backward(G)
    ref = torch.zeros(rows, width, dtype=torch.float64, device=d)
    ref.index_add_(0, idx[:, 0], G.double()); ref.index_add_(0, idx[:, 1], G.double())
    assert torch.allclose(table.grad.double(), ref, rtol=1e-5, atol=1e-5)
    with pytest.raises(IndexError):
        bad = idx.clone() if E else torch.zeros(1, 2, dtype=torch.long, device=d)
        bad[0, 1] = rows
        gat_layers.gather_rows_pair(table.detach(), bad)
