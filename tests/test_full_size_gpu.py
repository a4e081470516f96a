"""Parity at the FULL sizes BASELINE.json names, against the float64 oracle (not against another GEMM family of this
repository): every output and every gradient of the benchmarked configuration, cfg 4's per-GPU shard, the GP-GNN block form at
n = 32, GraphConvolution in bf16 at B = 1 024, and layer-by-layer gradient checks of the cfg 5 mixed bf16 stack.

Reference math: GAT/layers.py:111-178 (attention layer), models/models.py:240-274 (block adjacency + propagation),
models/layers.py:57-63 (GraphConvolution).  Tolerances: outputs 1e-4 absolute (north_star); gradients 1e-5 + 1e-4 of the
gradient's largest magnitude."""
import numpy as np
import pytest
import torch

from oracle import recon_oracle as O
from test_gat_gpu import close, dev, _power_law_batch

pytestmark = pytest.mark.gpu

_ORACLE_CACHE = {}


def _cfg2_problem(B, seed=0, copies=1):
    """`copies` identical sets of B graphs (node ids shifted): the batch the benchmark runs (copies = 1) or a batch whose halves
    must be bit-equal (copies = 2)."""
    n, e, F_, R, D, H = 16, 64, 200, 200, 200, 8
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, R, seed=seed)
    g = torch.Generator().manual_seed(0)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)])
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    G = torch.randn(B * n, H * D, generator=torch.Generator().manual_seed(1))
    if copies > 1:
        edge = torch.cat([edge + k * B * n for k in range(copies)], dim=1)
        x, ee, G = x.repeat(copies, 1), ee.repeat(copies, 1), G.repeat(copies, 1)
    return x, edge, ee, a, a2, G, (n, e, F_, R, D, H)


def _oracle_f64(key, x, edge, ee, a, a2, G, D, H):
    """All heads in float64 through the closed-form backward the oracle pins against the reference's autograd
    (tests/test_oracle_golden.py)."""
    if key not in _ORACLE_CACHE:
        out, g_x, g_ee, g_a, g_a2 = [], 0, 0, [], []
        for h in range(H):
            r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[h].double(), a2[h:h + 1].double(), 0.2, True,
                                     G[:, h * D:(h + 1) * D].double())
            out.append(r["out"]); g_a.append(r["g_a"]); g_a2.append(r["g_a_2"])
            g_x = g_x + r["g_x"]; g_ee = g_ee + r["g_edge_embed"]
        _ORACLE_CACHE.clear()                                  # one problem at a time: these are hundreds of MB
        _ORACLE_CACHE[key] = (torch.cat(out, 1), g_x, g_ee, torch.stack(g_a), torch.cat(g_a2))
    return _ORACLE_CACHE[key]


def _run_heads(x, edge, ee, a, a2, G):
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    d = dev()
    leaves = [t.to(d).requires_grad_(True) for t in (x, ee, a, a2)]
    out = gat_layers.gat_heads(*leaves, prepare_graph(edge.to(d), None, x.shape[0]), None, 0.2, True)
    grads = torch.autograd.grad(out, leaves, G.to(d))
    return [out.detach()] + [t.detach() for t in grads]


NAMES = ("out", "g_x", "g_edge_embed", "g_a", "g_a_2")


def test_full_size_cfg2_all_heads_and_gradients_vs_oracle(gemm_family):
    """BASELINE.json configs[1] exactly as bench.py runs it (512 graphs x 16 nodes x 64 edges, F = R = D = 200, 8 heads), once
    per GEMM family — "gemm_hx2" is the benchmarked one: `out` of all 8 heads, g_x, g_edge_embed, g_a (a sum over all 32 768
    edges through the split-K weight-gradient product) and g_a_2 against the float64 oracle."""
    x, edge, ee, a, a2, G, (n, e, F_, R, D, H) = _cfg2_problem(512)
    ref = _oracle_f64(("cfg2", 512), x, edge, ee, a, a2, G, D, H)
    got = _run_heads(x, edge, ee, a, a2, G)
    close(got[0], ref[0].float(), atol=1e-4, rel_to_max=0.0, what="cfg2 out")
    for name, u, v in zip(NAMES[1:], got[1:], ref[1:]):
        close(u, v.float().reshape(u.shape), atol=1e-5, rel_to_max=1e-4, what="cfg2 " + name)
    # the quantity the two-term representation is most exposed in: relative error of g_a per head, well inside fp32 round-off of a
    # 32 768-term sum
    rel = ((got[3].cpu().double() - ref[3]).abs().amax(dim=(1, 2)) / ref[3].abs().amax(dim=(1, 2))).max().item()
    assert rel < 2e-5, rel


def test_cfg4_per_gpu_shard_vs_oracle_and_bit_equal_halves():
    """BASELINE.json configs[3]: one GPU's shard of the 8 192-graph batch — 1 024 graphs, N = 16 384, E = 65 536 — built as two
    copies of 512 graphs.  Outputs and input gradients of the second copy must be BIT-equal to the first's (same per-graph work
    at other addresses), everything is checked against the float64 oracle, and the weight gradients are twice the one-copy sums."""
    x, edge, ee, a, a2, G, (n, e, F_, R, D, H) = _cfg2_problem(512, copies=2)
    assert x.shape[0] == 16384 and edge.shape[1] == 65536
    got = _run_heads(x, edge, ee, a, a2, G)
    Nh, Eh = 8192, 32768
    assert torch.equal(got[0][:Nh], got[0][Nh:]), "out: second half differs"
    assert torch.equal(got[1][:Nh], got[1][Nh:]), "g_x: second half differs"
    assert torch.equal(got[2][:Eh], got[2][Eh:]), "g_edge_embed: second half differs"
    x1, edge1, ee1, _, _, G1, _ = _cfg2_problem(512)
    ref = _oracle_f64(("cfg2", 512), x1, edge1, ee1, a, a2, G1, D, H)
    close(got[0][:Nh], ref[0].float(), atol=1e-4, rel_to_max=0.0, what="cfg4 shard out")
    close(got[1][:Nh], ref[1].float(), atol=1e-5, rel_to_max=1e-4, what="cfg4 shard g_x")
    close(got[2][:Eh], ref[2].float(), atol=1e-5, rel_to_max=1e-4, what="cfg4 shard g_edge_embed")
    close(got[3], 2 * ref[3].float(), atol=1e-5, rel_to_max=1e-4, what="cfg4 shard g_a")
    close(got[4], 2 * ref[4].float().reshape(got[4].shape), atol=1e-5, rel_to_max=1e-4, what="cfg4 shard g_a_2")


# ------------------------------------------------------------------------------- cfg 3b, GP-GNN block form at n = 32
def _prop_problem(n, d, L, B, seed, scale):
    from recon_amd.propagation import make_start_embedding, get_head_indices, get_tail_indices
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(seed)
    Ts = [torch.relu(torch.randn(B, C, dd * dd, generator=g)) * scale for _ in range(L)]
    ident = torch.eye(dd) + 0.02 * torch.randn(dd, dd, generator=g)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = torch.randn(B, C, S, 1, generator=g) * tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Gr = torch.randn(B, C, dd * L, generator=g)
    return Ts, ident, h0, head, tail, Gr


@pytest.mark.parametrize("act", ["relu", "tanh"])
def test_cfg3b_n32_forward_and_all_gradients_vs_oracle(act):
    """SURVEY 8d cfg 3b at n = 32 (S = 512, C = 992, 2d = 16, 3 hops untied, per-batch h0): the shape class the n = 9 tests never
    reach (16 column tiles per wave pass, 62 channel chunks per graph).  Forward and every gradient against the oracle in fp64."""
    from recon_amd.propagation import build_block_adjacency, propagate
    n, d, L, B = 32, 8, 3, 2
    Ts, ident, h0, head, tail, Gr = _prop_problem(n, d, L, B, seed=7, scale=0.02)

    def run(device, build, prop, dt, **kw):
        Tl = [t.clone().to(device=device, dtype=dt).requires_grad_(True) for t in Ts]
        I = ident.clone().to(device=device, dtype=dt).requires_grad_(True)
        h = h0.clone().to(device=device, dtype=dt).requires_grad_(True)
        out = prop([build(t, I, n) for t in Tl], h, act, head.to(device), tail.to(device), **kw)
        (out * Gr.to(device=device, dtype=dt)).sum().backward()
        return out.detach(), [t.grad for t in Tl], I.grad, h.grad
    # as_gemm: one matrix product per graph and hop instead of the reference's 1 984 broadcast matrix-vector products (6 minutes
    # and 17 GB in float64 at this size); pinned against the golden vectors in both forms by tests/test_oracle_golden.py
    out_r, gT_r, gI_r, gh_r = run("cpu", O.build_block_adjacency, O.propagate, torch.float64, as_gemm=True)
    out_h, gT_h, gI_h, gh_h = run(dev(), build_block_adjacency, propagate, torch.float32)
    close(out_h, out_r.float(), atol=1e-4, rel_to_max=1e-5, what="n32 out")
    close(gI_h, gI_r.float(), atol=1e-5, what="n32 g_identity")
    for l in range(L):
        close(gT_h[l], gT_r[l].float(), atol=1e-5, what="n32 g_T[%d]" % l)
    close(gh_h, gh_r.float(), atol=1e-5, what="n32 g_h0")


def test_cfg3b_n32_full_batch_properties():
    """cfg 3b at n = 32 with the full batch of 1 024 graphs (A_l = 1 GiB per hop, the channel states 2 GiB): (i) a slice against
    the oracle, (ii) graphs are independent (a sub-batch gives bit-equal rows), (iii) zero transition matrices under an identity
    diagonal leave the state unchanged."""
    from recon_amd.propagation import build_block_adjacency, propagate
    d_ = dev()
    n, d, L, B = 32, 8, 3, 1024
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    Ts, ident, h0, head, tail, _ = _prop_problem(n, d, L, 8, seed=3, scale=0.02)
    ident = torch.eye(dd)
    reps = B // 8
    adjs = [build_block_adjacency(t.to(d_).repeat(reps, 1, 1), ident.to(d_), n) for t in Ts]      # 128 copies of 8 graphs
    h0d = h0.to(d_).repeat(reps, 1, 1, 1)
    with torch.no_grad():
        out = propagate(adjs, h0d, "relu", head.to(d_), tail.to(d_))
    assert out.shape == (B, C, dd * L)
    ref = O.propagate([O.build_block_adjacency(t[:2].double(), ident.double(), n) for t in Ts], h0[:2].double(), "relu", head, tail, as_gemm=True)
    close(out[:2], ref.float(), atol=1e-4, rel_to_max=1e-5, what="cfg3b n32 slice")
    assert torch.equal(out[8:16], out[:8]) and torch.equal(out[B - 8:], out[:8]), "copies of the same graphs differ"
    with torch.no_grad():
        out2 = propagate([a[5:9].contiguous() for a in adjs], h0d[5:9].contiguous(), "relu", head.to(d_), tail.to(d_))
    assert torch.equal(out2, out[5:9])
    del adjs, out, out2
    zero = [build_block_adjacency(torch.zeros(4, C, dd * dd, device=d_), ident.to(d_), n) for _ in range(L)]
    h0p = h0[:4].abs().to(d_)
    with torch.no_grad():
        o3 = propagate(zero, h0p, "relu", head.to(d_), tail.to(d_))
    flat = h0p.view(4, C, S)
    expect = torch.gather(flat, 2, head.to(d_)[None].expand(4, -1, -1)) * torch.gather(flat, 2, tail.to(d_)[None].expand(4, -1, -1))
    close(o3, expect.repeat(1, 1, L), atol=1e-6, what="identity propagation at n = 32")


# ------------------------------------------------------------------------------- cfg 3a in bf16 at B = 1 024
def _bf(t):
    return t.to(torch.bfloat16)


def test_cfg3a_bf16_full_batch_vs_oracle():
    """BASELINE.json configs[2] as SURVEY 8d reads it (cfg 3a): B = 1 024 graphs, n = 32, D = 300, three GraphConvolution hops in
    bf16 storage / fp32 accumulation, the whole batch against the fp32 oracle on the same bf16-rounded operands, hop by hop
    (each hop's input is the kernels' own previous result, so what is compared is one layer's arithmetic: 2^-8 relative rounding
    of `support` and of the result), forward and every gradient of every hop."""
    from recon_amd.gcn_layers import GraphConvolution
    d_ = dev()
    B, n, D = 1024, 32, 300
    g = torch.Generator().manual_seed(9)
    x = _bf(torch.randn(B, n, D, generator=g))
    adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
    adj = _bf(adj / adj.sum(-1, keepdim=True))
    torch.manual_seed(3)
    layers = [GraphConvolution(D, D).to(torch.bfloat16).to(d_) for _ in range(3)]
    xd, adjd = x.to(d_).requires_grad_(True), adj.to(d_).requires_grad_(True)
    hs = [xd]
    for layer in layers:
        h = layer(hs[-1], adjd)
        h.retain_grad()
        hs.append(h)
    Gr = _bf(torch.randn(B, n, D, generator=g))
    (hs[-1] * Gr.to(d_)).sum().backward()
    adjf = adj.float()
    g_adj_sum = torch.zeros(B, n, n)
    for l, layer in enumerate(layers):
        w, b = layer.weight.detach().float().cpu(), layer.bias.detach().float().cpu()
        xin = hs[l].detach().float().cpu()
        out = hs[l + 1].detach().float().cpu()
        ref = O.graph_convolution(xin, adjf, w, b)
        close(out, ref, atol=1e-3, rel_to_max=1.5e-2, what="cfg3a bf16 hop %d out" % l)
        assert ((out > 0) != (ref > 0)).float().mean().item() < 0.02
        # models/layers.py:57-63 differentiated by hand, ReLU mask and upstream gradient of the bf16 run
        gup = hs[l + 1].grad.float().cpu()
        sup = _bf(xin @ w).float()
        gpre = gup * (out > 0)
        g_sup = _bf(adjf.transpose(1, 2) @ gpre).float()
        gin = hs[l].grad.float().cpu()
        close(gin, g_sup @ w.t(), atol=1e-3, rel_to_max=1e-2, what="cfg3a bf16 hop %d g_x" % l)
        close(layer.weight.grad.float().cpu(), xin.reshape(-1, D).t() @ g_sup.reshape(-1, D), atol=1e-3, rel_to_max=1e-2, what="hop %d g_weight" % l)
        close(layer.bias.grad.float().cpu(), gpre.reshape(-1, D).sum(0), atol=1e-3, rel_to_max=1e-2, what="hop %d g_bias" % l)
        g_adj_sum += gpre @ sup.transpose(1, 2)
    close(adjd.grad.float().cpu(), g_adj_sum, atol=2e-3, rel_to_max=2e-2, what="cfg3a bf16 g_adj (three hops)")


# ------------------------------------------------------------------------------- cfg 5: mixed bf16 stack, real gradient checks
def test_cfg5_bf16_stack_gradients_layer_by_layer():
    """BASELINE.json configs[4]: power-law graphs, an H-head attention layer with bf16 features in / out followed by three bf16
    GraphConvolutions on the same nodes.  Every layer's backward is checked against the oracle's gradient of THAT layer, given
    the layer's own (bf16-rounded) inputs and the upstream gradient the stack delivered to it: the attention layer through the
    float64 closed form (GAT/layers.py:111-178), the convolutions through models/layers.py:57-63 differentiated by hand."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    from recon_amd.gcn_layers import GraphConvolution
    d = dev()
    edge, N = _power_law_batch(4, seed=3, max_n=128)
    E = edge.shape[1]
    F_, R, D, H = 32, 16, 16, 4
    g = torch.Generator().manual_seed(0)
    x = _bf(torch.randn(N, F_, generator=g))
    ee = _bf(torch.randn(E, R, generator=g) * 0.5)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    xd, eed = x.to(d).requires_grad_(True), ee.to(d).requires_grad_(True)
    ad, a2d = a.to(d).requires_grad_(True), a2.to(d).requires_grad_(True)
    h = gat_layers.gat_heads(xd, eed, ad, a2d, prepare_graph(edge.to(d), None, N), None, 0.2, True)
    h.retain_grad()
    torch.manual_seed(2)
    adj = torch.zeros(N, N)
    adj[edge[0], edge[1]] = 1.0
    adj += torch.eye(N)
    adj = _bf(adj / adj.sum(-1, keepdim=True))
    layers = [GraphConvolution(H * D, H * D).to(torch.bfloat16).to(d) for _ in range(3)]
    hs = [h]
    for l in layers:
        cur = l(hs[-1], adj.to(d))
        cur.retain_grad()
        hs.append(cur)
    Gr = _bf(torch.randn(N, H * D, generator=g))
    (hs[-1] * Gr.to(d)).sum().backward()
    adjf = adj.float()
    for l, layer in enumerate(layers):
        w = layer.weight.detach().float().cpu()
        xin, out, gup = hs[l].detach().float().cpu(), hs[l + 1].detach().float().cpu(), hs[l + 1].grad.float().cpu()
        gpre = gup * (out > 0)
        g_sup = _bf(adjf.t() @ gpre).float()
        close(hs[l].grad.float().cpu(), g_sup @ w.t(), atol=1e-3, rel_to_max=1e-2, what="cfg5 conv %d g_x" % l)
        close(layer.weight.grad.float().cpu(), xin.t() @ g_sup, atol=1e-3, rel_to_max=1e-2, what="cfg5 conv %d g_weight" % l)
        close(layer.bias.grad.float().cpu(), gpre.sum(0), atol=1e-3, rel_to_max=1e-2, what="cfg5 conv %d g_bias" % l)
    # attention layer: upstream gradient = what the first convolution handed back (bf16), inputs = the bf16-rounded x / edge_embed
    gup = h.grad.float().cpu().double()
    g_x, g_ee = 0, 0
    for i in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[i].double(), a2[i:i + 1].double(), 0.2, True, gup[:, i * D:(i + 1) * D])
        g_x, g_ee = g_x + r["g_x"], g_ee + r["g_edge_embed"]
        close(ad.grad[i], r["g_a"].float(), atol=1e-5, rel_to_max=1e-4, what="cfg5 g_a head %d" % i)          # fp32 inside: fp32 bar
        close(a2d.grad[i], r["g_a_2"].float().reshape(-1), atol=1e-5, rel_to_max=1e-4, what="cfg5 g_a_2 head %d" % i)
    assert xd.grad.dtype == torch.bfloat16 and eed.grad.dtype == torch.bfloat16
    close(xd.grad.float(), g_x.float(), atol=1e-5, rel_to_max=2.0 ** -8, what="cfg5 g_x (bf16 out)")          # one rounding to bf16
    close(eed.grad.float(), g_ee.float(), atol=1e-5, rel_to_max=2.0 ** -8, what="cfg5 g_edge_embed (bf16 out)")


# ------------------------------------------------------------------------------- cfg 3b in bf16 at B = 1 024 (BASELINE.json configs[2])
@pytest.mark.parametrize("n,copies_of", [(9, 512), (32, 8)])
def test_cfg3b_bf16_full_batch_block_mode_training(n, copies_of):
    """The same configuration through propagate_blocks() with gradients — what a bfloat16 GPGNN / RECON_EAC training step runs: no
    adjacency is materialised, d T_l comes back in T's layout, d identity as one [16, 16] sum over graphs, nodes and hops.  (i) the first
    graphs (forward, d T, d h0) against the oracle from the forward's own states, (ii) d identity against the oracle's diagonal blocks
    summed over ALL graphs, (iii) copies bit-equal."""
    from recon_amd import propagation as P
    d_ = dev()
    d, L, B, act = 8, 3, 1024, "relu"
    C, S, dd = n * (n - 1), 16 * n, 16
    Ts, ident, h0, head, tail, Gr = _prop_problem(n, d, L, copies_of, seed=23, scale=1.2 / S ** 0.5)
    reps = B // copies_of
    Tb = [_bf(t).to(d_).repeat(reps, 1, 1).requires_grad_(True) for t in Ts]
    Ib = _bf(ident).to(d_).requires_grad_(True)
    hb = _bf(h0).to(d_).repeat(reps, 1, 1, 1).requires_grad_(True)
    Gb = _bf(Gr).to(d_).repeat(reps, 1, 1)
    out, states = P.propagate_blocks(Tb, Ib, n, hb, act, head.to(d_), tail.to(d_), return_states=True)
    assert states is not None and out.shape == (B, C, dd * L) and out.dtype == torch.bfloat16
    (out.float() * Gb.float()).sum().backward()
    k = copies_of if n <= 16 else 8                                    # graphs the oracle runs (all distinct ones: d identity sums over them)
    adj_r = [O.build_block_adjacency(_bf(t[:k]).float(), _bf(ident).float(), n) for t in Ts]
    h0_r = _bf(h0[:k]).float()
    ref = O.propagate(adj_r, h0_r, act, head, tail, as_gemm=True, storage=torch.bfloat16)
    close(out[:k].float(), ref, atol=1e-3, rel_to_max=1.5e-2, what="blocks bf16 n=%d out" % n)
    g_adj_r, g_h_r = O.propagate_backward(adj_r, h0_r, [s_[:k].float().cpu() for s_ in states], act, head, tail, _bf(Gr[:k]).float(), storage=torch.bfloat16)
    gI = torch.zeros(dd, dd)
    for l in range(L):
        blocks = g_adj_r[l].reshape(k, n, dd, n, dd).permute(0, 1, 3, 2, 4)
        off = torch.stack([blocks[:, i, j] for i in range(n) for j in range(n) if i != j], 1).reshape(k, C, dd * dd)
        close(Tb[l].grad[:k].float(), off, atol=1e-3, rel_to_max=2e-2, what="blocks bf16 n=%d g_T[%d]" % (n, l))
        gI += torch.stack([blocks[:, i, i] for i in range(n)], 1).sum((0, 1))
    close(hb.grad[:k].float().reshape(g_h_r.shape), g_h_r, atol=1e-3, rel_to_max=2e-2, what="blocks bf16 n=%d g_h0" % n)
    close(Ib.grad.float(), gI * (B // k), atol=1e-2, rel_to_max=2e-2, what="blocks bf16 n=%d g_identity" % n)
    for t in [out, hb.grad] + [t.grad for t in Tb]:
        v = t.view(reps, copies_of, -1)
        assert torch.equal(v[1], v[0]) and torch.equal(v[reps - 1], v[0]), "copies of the same graphs differ"


@pytest.mark.parametrize("n,copies_of", [(9, 512), (32, 8)])
def test_cfg3b_bf16_full_batch_forward_and_all_gradients(n, copies_of):
    """BASELINE.json configs[2] "GP-GNN Propagation 3 hops, batch 1024 graphs, 32 nodes ... bf16" (and the reference's own n = 9): block
    adjacency + three hops + every gradient on bf16 tensors at B = 1 024.  The batch is 1 024 / copies_of copies of `copies_of` graphs:
    (i) the first graphs against the fp32 oracle on the same bf16 operands, states rounded hop by hop, gradients from the forward's own
    states (tests/test_prop_b16_gpu.py has the method), (ii) every copy bit-equal to the first (graphs are independent, the kernels
    deterministic)."""
    from recon_amd import propagation as P
    d_ = dev()
    d, L, B, act = 8, 3, 1024, "relu"
    C, S, dd = n * (n - 1), 16 * n, 16
    Ts, ident, h0, head, tail, Gr = _prop_problem(n, d, L, copies_of, seed=21, scale=1.2 / S ** 0.5)
    reps = B // copies_of
    Tb = [_bf(t).to(d_).repeat(reps, 1, 1).requires_grad_(True) for t in Ts]
    Ib = _bf(ident).to(d_).requires_grad_(True)
    hb = _bf(h0).to(d_).repeat(reps, 1, 1, 1).requires_grad_(True)
    Gb = _bf(Gr).to(d_).repeat(reps, 1, 1)
    adjs = [P.build_block_adjacency(t, Ib, n) for t in Tb]
    for a in adjs:
        a.retain_grad()
    out, states = P.propagate(adjs, hb, act, head.to(d_), tail.to(d_), return_states=True)
    assert out.shape == (B, C, dd * L) and out.dtype == torch.bfloat16
    (out.float() * Gb.float()).sum().backward()
    k = copies_of if n <= 16 else 8                                    # graphs the oracle runs: all distinct ones (d identity sums over them)
    adj_r = [O.build_block_adjacency(_bf(t[:k]).float(), _bf(ident).float(), n) for t in Ts]
    h0_r = _bf(h0[:k]).float()
    ref = O.propagate(adj_r, h0_r, act, head, tail, as_gemm=True, storage=torch.bfloat16)
    close(out[:k].float(), ref, atol=1e-3, rel_to_max=1.5e-2, what="cfg3b bf16 n=%d out" % n)
    g_adj_r, g_h_r = O.propagate_backward(adj_r, h0_r, [s[:k].float().cpu() for s in states], act, head, tail, _bf(Gr[:k]).float(), storage=torch.bfloat16)
    gI = torch.zeros(dd, dd)
    for l in range(L):
        gI += torch.stack([g_adj_r[l].reshape(k, n, dd, n, dd)[:, i, :, i, :] for i in range(n)], 1).sum((0, 1))
        close(adjs[l].grad[:k].float(), g_adj_r[l], atol=1e-3, rel_to_max=2e-2, what="cfg3b bf16 n=%d g_adj[%d]" % (n, l))
        blocks = adjs[l].grad[:k].float().cpu().reshape(k, n, dd, n, dd).permute(0, 1, 3, 2, 4)
        off = torch.stack([blocks[:, i, j] for i in range(n) for j in range(n) if i != j], 1).reshape(k, C, dd * dd)
        assert torch.equal(Tb[l].grad[:k].float().cpu(), off), "d T is the off-diagonal blocks of d A"
    close(hb.grad[:k].float().reshape(g_h_r.shape), g_h_r, atol=1e-3, rel_to_max=2e-2, what="cfg3b bf16 n=%d g_h0" % n)
    close(Ib.grad.float(), gI * (B // k), atol=1e-2, rel_to_max=2e-2, what="cfg3b bf16 n=%d g_identity (adjacency form, sum over all graphs)" % n)
    for t in [out, hb.grad] + [a.grad for a in adjs]:
        v = t.view(reps, copies_of, -1)
        assert torch.equal(v[1], v[0]) and torch.equal(v[reps - 1], v[0]), "copies of the same graphs differ"


def test_cfg5_bf16_ragged_stack_at_256_nodes():
    """BASELINE.json configs[4] as stated: power-law graphs of up to 256 nodes / 4 096 edges each, an H-head attention layer with bf16
    features in / out, then three bf16 GraphConvolutions applied PER GRAPH (every graph its own row-normalised dense adjacency: a ragged
    batch, recon_amd.gcn_layers.RaggedAdjacency).  Every layer's backward against the oracle's gradient of that layer, given the layer's
    own bf16 inputs and the upstream gradient the stack delivered (GAT/layers.py:111-178 in float64; models/layers.py:57-63 by hand)."""
    from recon_amd import gat_layers
    from recon_amd.graph import prepare_graph
    from recon_amd.gcn_layers import GraphConvolution, RaggedAdjacency
    d = dev()
    rs = np.random.RandomState(5)
    sizes = [256, 256] + [int(v) for v in rs.randint(16, 257, size=10)]
    dsts, srcs, mats, base = [], [], [], 0
    for n in sizes:                                                     # SURVEY 8d cfg-5 generator, sizes fixed so that two graphs have 256 nodes
        e = min(4096, 16 * n)
        p = 1.0 / np.arange(1, n + 1)
        p /= p.sum()
        dl, sl = rs.choice(n, size=e, p=p), rs.randint(0, n, size=e)
        a = torch.zeros(n, n)
        a[torch.from_numpy(dl), torch.from_numpy(sl)] = 1.0
        a += torch.eye(n)
        mats.append(_bf(a / a.sum(-1, keepdim=True)))
        dsts.append(dl + base); srcs.append(sl + base)
        base += n
    N = base
    edge = torch.from_numpy(np.stack([np.concatenate(dsts), np.concatenate(srcs)])).long()
    E = edge.shape[1]
    assert max(sizes) == 256 and E >= 2 * 4096
    F_, R, D, H = 32, 16, 16, 4
    g = torch.Generator().manual_seed(0)
    x = _bf(torch.randn(N, F_, generator=g))
    ee = _bf(torch.randn(E, R, generator=g) * 0.5)
    a = torch.stack([O.xavier_normal((D, 2 * F_ + R), 1.414, g) for _ in range(H)]) * 0.5
    a2 = torch.cat([O.xavier_normal((1, D), 1.414, g) for _ in range(H)])
    xd, eed = x.to(d).requires_grad_(True), ee.to(d).requires_grad_(True)
    ad, a2d = a.to(d).requires_grad_(True), a2.to(d).requires_grad_(True)
    h = gat_layers.gat_heads(xd, eed, ad, a2d, prepare_graph(edge.to(d), None, N), None, 0.2, True)
    h.retain_grad()
    torch.manual_seed(2)
    rag = RaggedAdjacency.from_dense([m.to(d) for m in mats])
    layers = [GraphConvolution(H * D, H * D).to(torch.bfloat16).to(d) for _ in range(3)]
    hs = [h]
    for l in layers:
        cur = l(hs[-1], rag)
        cur.retain_grad()
        hs.append(cur)
    Gr = _bf(torch.randn(N, H * D, generator=g))
    (hs[-1] * Gr.to(d)).sum().backward()
    for l, layer in enumerate(layers):
        w, b = layer.weight.detach().float().cpu(), layer.bias.detach().float().cpu()
        xin, out, gup = hs[l].detach().float().cpu(), hs[l + 1].detach().float().cpu(), hs[l + 1].grad.float().cpu()
        gx_ref, gw_ref, gb_ref, r0 = torch.zeros_like(xin), torch.zeros_like(w), torch.zeros_like(b), 0
        for n, m in zip(sizes, mats):
            mf = m.float()
            close(out[r0:r0 + n], O.graph_convolution(xin[r0:r0 + n], mf, w, b), atol=1e-3, rel_to_max=1.5e-2, what="cfg5 ragged conv %d out (n=%d)" % (l, n))
            gpre = gup[r0:r0 + n] * (out[r0:r0 + n] > 0)
            g_sup = _bf(mf.t() @ gpre).float()
            gx_ref[r0:r0 + n] = g_sup @ w.t()
            gw_ref += xin[r0:r0 + n].t() @ g_sup
            gb_ref += gpre.sum(0)
            r0 += n
        close(hs[l].grad.float().cpu(), gx_ref, atol=1e-3, rel_to_max=1e-2, what="cfg5 ragged conv %d g_x" % l)
        close(layer.weight.grad.float().cpu(), gw_ref, atol=1e-3, rel_to_max=1e-2, what="cfg5 ragged conv %d g_weight" % l)
        close(layer.bias.grad.float().cpu(), gb_ref, atol=1e-3, rel_to_max=1e-2, what="cfg5 ragged conv %d g_bias" % l)
    gup = h.grad.float().cpu().double()
    g_x = 0
    for hh in range(H):
        r = O.gat_layer_backward(x.double(), edge, ee.double(), None, None, a[hh].double(), a2[hh:hh + 1].double(), 0.2, True, gup[:, hh * D:(hh + 1) * D])
        close(h.detach().float().cpu()[:, hh * D:(hh + 1) * D], r["out"].float(), atol=1e-3, rel_to_max=1e-2, what="cfg5 ragged gat out h%d" % hh)
        g_x = g_x + r["g_x"]
    close(xd.grad.float().cpu(), g_x.float(), atol=1e-3, rel_to_max=1e-2, what="cfg5 ragged gat g_x")


# ------------------------------------------------------------------------------- cfg 3b at n = 32 in float32: the backward at the full batch
@pytest.mark.parametrize("B,copies_of", [(300, 6), (1024, 8)])
def test_cfg3b_n32_fp32_backward_across_slices_and_full_batch(B, copies_of):
    """The wide-state backward (both products of a hop as batched GEMMs, csrc/prop.hip prop_bwd_wide) where round 3 had no gradient check:
    B = 300 crosses a slice boundary of the forward's split workspace (256 graphs per slice), B = 1 024 is the benchmarked batch.  The
    batch is copies of `copies_of` graphs: (i) the distinct graphs — forward and EVERY gradient (d T_l, d h0, and d identity, which sums the
    diagonal blocks over ALL graphs: the oracle's sum over the distinct ones times the number of copies) — against the float64 oracle,
    (ii) every copy bit-equal to the first (graphs are independent, the kernels deterministic), which covers the graphs past the slice boundary."""
    from recon_amd.propagation import build_block_adjacency, propagate
    d_ = dev()
    n, d, L, act = 32, 8, 3, "relu"
    C, S, dd = n * (n - 1), 16 * n, 16
    Ts, ident, h0, head, tail, Gr = _prop_problem(n, d, L, copies_of, seed=31, scale=0.02)
    reps = B // copies_of
    Tb = [t.to(d_).repeat(reps, 1, 1).requires_grad_(True) for t in Ts]
    Ib = ident.to(d_).requires_grad_(True)
    hb = h0.to(d_).repeat(reps, 1, 1, 1).requires_grad_(True)
    Gb = Gr.to(d_).repeat(reps, 1, 1)
    adjs = [build_block_adjacency(t, Ib, n) for t in Tb]
    out = propagate(adjs, hb, act, head.to(d_), tail.to(d_))
    (out * Gb).sum().backward()
    del adjs
    k = copies_of
    Tr = [t[:k].double().requires_grad_(True) for t in Ts]
    Ir = ident.double().requires_grad_(True)
    hr = h0[:k].double().requires_grad_(True)
    ref = O.propagate([O.build_block_adjacency(t, Ir, n) for t in Tr], hr, act, head, tail, as_gemm=True)
    (ref * Gr[:k].double()).sum().backward()
    close(out[:k], ref.detach().float(), atol=1e-4, rel_to_max=1e-5, what="n32 fp32 B=%d out" % B)
    for l in range(L):
        close(Tb[l].grad[:k], Tr[l].grad.float(), atol=1e-5, what="n32 fp32 B=%d g_T[%d]" % (B, l))
    close(hb.grad[:k], hr.grad.float(), atol=1e-5, what="n32 fp32 B=%d g_h0" % B)
    close(Ib.grad, (Ir.grad * reps).float(), atol=1e-4, rel_to_max=2e-5, what="n32 fp32 B=%d g_identity (sum over all %d graphs)" % (B, B))
    for t in [out, hb.grad] + [t.grad for t in Tb]:
        v = t.reshape(reps, copies_of, -1)
        assert not bool((v != v[:1]).any()), "copies of the same graphs differ"          # EVERY copy, not a sample: a race shows up in a few workgroups only


def test_propagate_blocks_trains_at_n32():
    """propagate_blocks() with gradients at n = 32: the forward reads the transition tensors in place, the backward's chain and d T products
    run on the two-term f16 kernels of csrc/prop_hl.hip (transposed split straight from T, d T written in T's layout, diagonal blocks summed
    into d identity) — no adjacency in either direction.  Forward and every gradient against the float64 oracle."""
    from recon_amd.propagation import propagate_blocks, _blocks_wide_trainable
    d_ = dev()
    assert _blocks_wide_trainable(3, 32, 16, torch.zeros(3, 992, 512, 1, device=d_), 2, _prop_problem(32, 8, 1, 1, 0, 1.0)[3], _prop_problem(32, 8, 1, 1, 0, 1.0)[4])
    n, d, L, B, act = 32, 8, 2, 3, "tanh"
    Ts, ident, h0, head, tail, Gr = _prop_problem(n, d, L, B, seed=17, scale=0.02)
    Tl = [t.to(d_).requires_grad_(True) for t in Ts]
    I = ident.to(d_).requires_grad_(True)
    h = h0.to(d_).requires_grad_(True)
    out = propagate_blocks(Tl, I, n, h, act, head.to(d_), tail.to(d_))
    (out * Gr.to(d_)).sum().backward()
    Tr = [t.double().requires_grad_(True) for t in Ts]
    Ir, hr = ident.double().requires_grad_(True), h0.double().requires_grad_(True)
    ref = O.propagate([O.build_block_adjacency(t, Ir, n) for t in Tr], hr, act, head, tail, as_gemm=True)
    (ref * Gr.double()).sum().backward()
    close(out, ref.detach().float(), atol=1e-4, rel_to_max=1e-5, what="blocks n32 training out")
    close(I.grad, Ir.grad.float(), atol=1e-5, what="blocks n32 g_identity")
    close(h.grad, hr.grad.float(), atol=1e-5, what="blocks n32 g_h0")
    for l in range(L):
        close(Tl[l].grad, Tr[l].grad.float(), atol=1e-5, what="blocks n32 g_T[%d]" % l)
