"""GPU parity tests for the GP-GNN side: block adjacency, L-hop propagation (+ head*tail gather),
start-entity embeddings and GraphConvolution — HIP kernels vs golden vectors from the reference and
vs the CPU oracle.  fp32 tolerance 1e-4 abs on outputs (north_star)."""
import numpy as np
import pytest
import torch

from conftest import load_golden, hashed_uniform
from oracle import recon_oracle as O
from test_gat_gpu import close, dev

pytestmark = pytest.mark.gpu
T = torch.from_numpy


def _prop_inputs(g):
    n, d, L, B, salt = (int(g[k]) for k in ("n", "d", "L", "B", "salt"))
    C, S, dd = n * (n - 1), 2 * d * n, (2 * d) ** 2
    Ts = [T(hashed_uniform((B, C, dd), salt * 10 + i, -0.6, 1.0)) for i in range(L)]
    per_batch = "g_h0_sum" in g
    h0 = T(g["h0_shared"])
    if per_batch:
        h0 = T(hashed_uniform((B, C, S, 1), salt * 10 + 8)) * h0
    Gr = T(hashed_uniform(g["out"].shape, int(g["G_salt"])))
    return n, d, L, B, Ts, h0, Gr, per_batch


@pytest.mark.parametrize("name", ["prop_n4d2_shared", "prop_n4d2_perbatch", "prop_n9d8_shared", "prop_n9d8_perbatch"])
def test_propagation_golden(name):
    """The reference's GPGNN untied branch (models/models.py:238-277), replayed through the drop-in
    functions: relu -> build_block_adjacency -> propagate, outputs and all gradients."""
    from recon_amd.propagation import build_block_adjacency, propagate
    g = load_golden(name)
    d_ = dev()
    n, d, L, B, Ts, h0, Gr, per_batch = _prop_inputs(g)
    Ts = [t.to(d_).requires_grad_(True) for t in Ts]
    ident = T(g["identity"]).to(d_).requires_grad_(True)
    h0 = h0.to(d_).requires_grad_(per_batch)
    adjs = [build_block_adjacency(torch.relu(t), ident, n) for t in Ts]
    for l in range(L):
        np.testing.assert_array_equal(adjs[l][0].detach().cpu().numpy(), g["adj_b0"][l])
        np.testing.assert_array_equal(adjs[l][B - 1].detach().cpu().numpy(), g["adj_bl"][l])
    head = T(np.tile(g["head_indices"][None], (B, 1, 1))).to(d_)       # the reference stores [bs, C, 2d]
    tail = T(np.tile(g["tail_indices"][None], (B, 1, 1))).to(d_)
    out = propagate([a.view(B, 1, a.shape[1], a.shape[2]) for a in adjs], h0, "relu", head, tail)
    close(out, g["out"], what=name + " out")
    (out * Gr.to(d_)).sum().backward()
    close(ident.grad, g["g_identity"], atol=1e-4, what="g_identity")
    for l in range(L):
        close(Ts[l].grad[0], g["g_T_b0"][l], atol=1e-5, what="g_T[%d][0]" % l)
        close(Ts[l].grad.sum(0), g["g_T_sum"][l], atol=1e-4, what="sum_b g_T[%d]" % l)
        if "g_T" in g:
            close(Ts[l].grad, g["g_T"][l], atol=1e-5, what="g_T[%d]" % l)
    if per_batch:
        close(h0.grad[0], g["g_h0_b0"], atol=1e-5, what="g_h0[0]")
        close(h0.grad.sum(0), g["g_h0_sum"], atol=1e-4, what="sum_b g_h0")


def test_propagation_beyond_4_gib_per_hop():
    """Maximum sizes: B = 60 000 graphs of cfg 3b's shape (n = 9, d = 8, 3 hops) — every A_l is 1.24 G elements = 4.98 GB, past
    32-bit byte offsets; the batch limit of the kernels is 65 535.  The batch is two copies of 30 000 graphs: both halves of the
    outputs and of every gradient must be bit-equal, and the first graphs are checked against the oracle."""
    from recon_amd.propagation import build_block_adjacency, propagate, make_start_embedding, get_head_indices, get_tail_indices
    d_ = dev()
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~70 GB of device memory")
    n, d, L, Bh = 9, 8, 3, 30000
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(5)
    Th = [(torch.rand(Bh, C, dd * dd, generator=g) - 0.4) * 0.2 for _ in range(L)]
    ident0 = torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0h = torch.randn(Bh, C, S, 1, generator=g) * tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Grh = torch.randn(Bh, C, dd * L, generator=g)
    ident = ident0.clone().to(d_).requires_grad_(True)
    Tl = [torch.cat([t, t]).to(d_).requires_grad_(True) for t in Th]
    h = torch.cat([h0h, h0h]).to(d_).requires_grad_(True)
    adjs = [build_block_adjacency(torch.relu(t), ident, n) for t in Tl]
    assert adjs[0].numel() * 4 > 2 ** 32
    out = propagate(adjs, h, "relu", head.to(d_), tail.to(d_))
    (out * torch.cat([Grh, Grh]).to(d_)).sum().backward()
    assert torch.equal(out[:Bh], out[Bh:])
    assert torch.equal(h.grad[:Bh], h.grad[Bh:])
    for l in range(L):
        assert torch.equal(Tl[l].grad[:Bh], Tl[l].grad[Bh:])
    k = 3
    Ts = [t[:k].clone().requires_grad_(True) for t in Th]
    hh = h0h[:k].clone().requires_grad_(True)
    I = ident0.clone().requires_grad_(True)
    o = O.propagate([O.build_block_adjacency(torch.relu(t), I, n) for t in Ts], hh, "relu", head, tail)
    (o * Grh[:k]).sum().backward()
    close(out[:k], o, what="out")
    close(h.grad[:k], hh.grad, what="g_h0")
    for l in range(L):
        close(Tl[l].grad[:k], Ts[l].grad, what="g_T[%d]" % l)


@pytest.mark.parametrize("n,d,L,B,act,per_batch", [
    (3, 2, 1, 2, "relu", False),       # S = 12: padded to one 16-wide MFMA tile
    (5, 3, 3, 3, "tanh", True),        # S = 30: not a multiple of 4 -> scalar loads
    (9, 8, 3, 5, "linear", True),      # model_params.json sizes
    (6, 4, 4, 1, "relu", True),        # B = 1, 4 hops
    (12, 4, 2, 2, "relu", False),      # C = 132 > 80: two channel chunks per graph
])
@pytest.mark.parametrize("form", ["f16x2", "bf16x3", "wave", "block"])
def test_propagation_vs_oracle(n, d, L, B, act, per_batch, form, recon_config):
    """`form`: the forward kernel runs on the f16 matrix cores with two-term operands (the default where S % 4 == 0, S <= 160,
    C <= 96), on the fp32 matrix cores with the channel states in registers per wave (S <= 144) / in LDS per workgroup, or
    (opt-in) on the bf16 matrix cores with three-term split operands."""
    recon_config("RECON_PROP_FWD", {"f16x2": "h", "block": "b", "wave": "w", "bf16x3": "x"}[form])
    from recon_amd.propagation import build_block_adjacency, propagate, make_start_embedding, get_head_indices, get_tail_indices
    d_ = dev()
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(n * 100 + d)
    Ts = [(torch.rand(B, C, dd * dd, generator=g) - 0.4) * 0.5 for _ in range(L)]
    ident = torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(B, C, S, 1, generator=g) * tmpl) if per_batch else tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Gr = torch.randn(B, C, dd * L, generator=g)

    def run(device, build, prop):
        Tl = [t.clone().to(device).requires_grad_(True) for t in Ts]
        I = ident.clone().to(device).requires_grad_(True)
        h = h0.clone().to(device).requires_grad_(per_batch)
        adjs = [build(torch.relu(t), I, n) for t in Tl]
        out = prop(adjs, h, act, head.to(device), tail.to(device))
        (out * Gr.to(device)).sum().backward()
        return out, [t.grad for t in Tl], I.grad, h.grad
    out_r, gT_r, gI_r, gh_r = run("cpu", O.build_block_adjacency, O.propagate)
    out_h, gT_h, gI_h, gh_h = run(d_, build_block_adjacency, propagate)
    close(out_h, out_r, what="out")
    close(gI_h, gI_r, atol=1e-5, what="g_identity")
    for l in range(L):
        close(gT_h[l], gT_r[l], atol=1e-5, what="g_T[%d]" % l)
    if per_batch:
        close(gh_h, gh_r, atol=1e-5, what="g_h0")


@pytest.mark.parametrize("S,C,dd,L,B,act,per_batch,grad", [
    (512, 130, 16, 3, 3, "relu", True, False),      # three chunks, the last one of 2 channels; inference
    (512, 64, 16, 2, 2, "tanh", False, False),      # exactly one chunk, shared h0
    (176, 5, 8, 3, 4, "linear", True, False),       # RT = 2, K padded from 6 to 6 steps ... 176 = 5.5 steps -> 6
    (272, 70, 16, 2, 9, "relu", True, True),        # RT = 3, odd K steps padded to even; states saved, fp32 backward
    (400, 33, 20, 1, 2, "relu", True, False),       # gather width 20: more than two items per thread
    (256, 96, 4, 3, 11, "tanh", True, True),        # RT = 2 exactly; more than 8 graphs: two XCD rounds
    (164, 1, 2, 2, 1, "relu", True, False),         # S % 16 != 0: a partial row tile and a partial K step; one channel, one graph
    (192, 3, 8, 2, 300, "relu", False, False),      # more graphs than one slice of the split workspace (256): two split + propagation rounds
    (512, 3, 4, 8, 1, "tanh", False, False),        # eight hops at S = 512: the two-term form's LDS image does not fit, the fp32 form answers
])
def test_propagation_wide_states_vs_oracle(S, C, dd, L, B, act, per_batch, grad):
    """160 < S <= 512: the two-term f16 form of csrc/prop_hl.hip (A_l pre-split per slice of graphs, 64-channel chunks per workgroup)
    against the float64 oracle, on arbitrary adjacencies, start states and gather indices; with gradients the states it saves feed
    the fp32 backward."""
    from recon_amd.propagation import propagate
    d_ = dev()
    g = torch.Generator().manual_seed(S + C)
    adjs = [(torch.rand(B, S, S, generator=g) - 0.45) * (2.0 / S ** 0.5) * (1 + l) for l in range(L)]
    for a in adjs:
        a[:, :, ::7] *= 8.0                                            # columns of very different magnitude inside a row
        a[:, 5] *= 1e-3                                                # and rows: every row has its own scale
    h0 = torch.randn(B, C, S, 1, generator=g) if per_batch else torch.randn(C, S, 1, generator=g)
    head = torch.randint(0, S, (C, dd), generator=g)
    tail = torch.randint(0, S, (C, dd), generator=g)
    Gr = torch.randn(B, C, dd * L, generator=g)

    def run(device, prop, dt):
        A = [a.clone().to(device=device, dtype=dt).requires_grad_(grad) for a in adjs]
        h = h0.clone().to(device=device, dtype=dt).requires_grad_(grad and per_batch)
        out = prop(A, h, act, head.to(device), tail.to(device))
        if grad:
            (out * Gr.to(device=device, dtype=dt)).sum().backward()
        return out.detach(), [a.grad for a in A], h.grad
    out_r, gA_r, gh_r = run("cpu", lambda *a: O.propagate(*a, as_gemm=True), torch.float64)
    out_h, gA_h, gh_h = run(d_, propagate, torch.float32)
    close(out_h, out_r.float(), atol=1e-4, rel_to_max=1e-5, what="out")
    if grad:
        for l in range(L):
            close(gA_h[l], gA_r[l].float(), atol=1e-5, what="g_adj[%d]" % l)
        if per_batch:
            close(gh_h, gh_r.float(), atol=1e-5, what="g_h0")


@pytest.mark.parametrize("n,L,B,act", [(11, 3, 3, "relu"), (32, 3, 2, "relu"), (17, 2, 9, "tanh")])
def test_propagate_blocks_wide_states_inference(n, L, B, act):
    """Block mode for 10 < n <= 32 (inference): the split pass reads the transition tensors and the identity in place — the S x S
    adjacency is never built — and must agree bit for bit with the same kernels fed the materialised adjacency, and with the oracle."""
    from recon_amd.propagation import (build_block_adjacency, propagate, propagate_blocks, blocks_mode_available, make_start_embedding,
                                       get_head_indices, get_tail_indices)
    d_ = dev()
    d = 8
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(n)
    Ts = [torch.relu(torch.randn(B, C, dd * dd, generator=g)) * (0.5 / n) for _ in range(L)]
    ident = torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = torch.randn(B, C, S, 1, generator=g) * tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    with torch.no_grad():
        assert blocks_mode_available(B, n, dd, h0.to(d_), need_grad=False)
        assert not blocks_mode_available(B, n, dd, h0.to(d_), need_grad=True)
        Td, Id, hd = [t.to(d_) for t in Ts], ident.to(d_), h0.to(d_)
        out_f = propagate_blocks(Td, Id, n, hd, act, head.to(d_), tail.to(d_))
        out_u = propagate([build_block_adjacency(t, Id, n) for t in Td], hd, act, head.to(d_), tail.to(d_))
        ref = O.propagate([O.build_block_adjacency(t.double(), ident.double(), n) for t in Ts], h0.double(), act, head, tail, as_gemm=True)
    assert torch.equal(out_f, out_u)
    close(out_f, ref.float(), atol=1e-4, rel_to_max=1e-5, what="out")


@pytest.mark.parametrize("n,L,B,act,per_batch,tied", [(9, 3, 7, "relu", True, False), (9, 3, 300, "relu", False, False), (4, 2, 5, "tanh", True, False),
                                                       (10, 1, 3, "linear", True, False), (6, 3, 4, "relu", True, True)])
def test_propagate_blocks_matches_unfused(n, L, B, act, per_batch, tied):
    """propagate_blocks (models/models.py:240-274 in one call: the kernels read A_l out of the transition tensors in place and write
    d loss / d T in T's layout) against build_block_adjacency + propagate: the SAME arithmetic on the same values — forward, d T and
    d h0 bit-equal; d identity to round-off (per-workgroup partial sums instead of the stand-alone reduction) — and against the oracle."""
    from recon_amd.propagation import (build_block_adjacency, propagate, propagate_blocks, blocks_mode_available, make_start_embedding,
                                       get_head_indices, get_tail_indices)
    d_ = dev()
    d = 8
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(n * 10 + L)
    Ts = [torch.relu(torch.randn(B, C, dd * dd, generator=g)) * 0.1 for _ in range(1 if tied else L)]
    ident = torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(B, C, S, 1, generator=g) * tmpl) if per_batch else tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Gr = torch.randn(B, C, dd * L, generator=g)
    assert blocks_mode_available(B, n, dd, h0.to(d_)) == (n <= 9)          # n = 10: the backward's LDS image does not fit: unfused path, same answers

    def run(device, fused, dt=torch.float32):
        Tl = [t.clone().to(device=device, dtype=dt).requires_grad_(True) for t in Ts]
        tl = Tl * L if tied else Tl
        I = ident.clone().to(device=device, dtype=dt).requires_grad_(True)
        h = h0.clone().to(device=device, dtype=dt).requires_grad_(per_batch)
        if fused:
            out = propagate_blocks(tl, I, n, h, act, head.to(device), tail.to(device))
        elif device == "cpu":
            out = O.propagate([O.build_block_adjacency(t, I, n) for t in tl], h, act, head, tail)
        else:
            out = propagate([build_block_adjacency(t, I, n) for t in tl], h, act, head.to(device), tail.to(device))
        (out * Gr.to(device=device, dtype=dt)).sum().backward()
        return out.detach(), [t.grad for t in Tl], I.grad, h.grad
    out_f, gT_f, gI_f, gh_f = run(d_, True)
    out_u, gT_u, gI_u, gh_u = run(d_, False)
    assert torch.equal(out_f, out_u)
    for a, b in zip(gT_f, gT_u):
        if tied:
            close(a, b, atol=1e-6, rel_to_max=1e-6, what="g_T (tied: L gradients summed by autograd vs in one buffer)")
        else:
            assert torch.equal(a, b)
    if per_batch:
        assert torch.equal(gh_f, gh_u)
    close(gI_f, gI_u, atol=1e-5, rel_to_max=1e-5, what="g_identity fused vs unfused")
    out_r, gT_r, gI_r, gh_r = run("cpu", False, torch.float64)
    close(out_f, out_r.float(), what="out vs oracle")
    close(gI_f, gI_r.float(), atol=1e-5, what="g_identity vs oracle")
    for a, b in zip(gT_f, gT_r):
        close(a, b.float(), atol=1e-5, what="g_T vs oracle")


def test_propagate_blocks_many_hops_at_wide_states_falls_back():
    """n = 32 with eight hops: the wide two-term form's LDS image does not fit that many hops (its budget grows with L), so propagate_blocks()
    must answer through block adjacency + propagate instead of raising (advisor, round 3: the availability probe used L = 1)."""
    from recon_amd.propagation import propagate_blocks, blocks_mode_available, make_start_embedding, get_head_indices, get_tail_indices
    d_ = dev()
    n, d, L, B = 32, 8, 8, 1
    dd, C, S = 16, n * (n - 1), 16 * n
    g = torch.Generator().manual_seed(2)
    Ts = [torch.relu(torch.randn(B, C, dd * dd, generator=g)) * 0.01 for _ in range(L)]
    ident = torch.eye(dd)
    h0 = torch.randn(B, C, S, 1, generator=g) * torch.from_numpy(make_start_embedding(n, d)).float()
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    assert blocks_mode_available(B, n, dd, h0.to(d_), need_grad=False, L=1) and not blocks_mode_available(B, n, dd, h0.to(d_), need_grad=False, L=L)
    with torch.no_grad():
        out = propagate_blocks([t.to(d_) for t in Ts], ident.to(d_), n, h0.to(d_), "tanh", head.to(d_), tail.to(d_))
    ref = O.propagate([O.build_block_adjacency(t.double(), ident.double(), n) for t in Ts], h0.double(), "tanh", head, tail, as_gemm=True)
    close(out, ref.float(), atol=1e-4, rel_to_max=1e-5, what="blocks n=32 L=8")


def test_start_entity_embeddings_golden():
    from recon_amd.propagation import make_start_entity_embeddings
    g = load_golden("prop3_start_entity")
    d_ = dev()
    ent = T(g["entity_embeddings"]).to(d_).requires_grad_(True)
    out = make_start_entity_embeddings(ent, T(g["pos"]).to(d_), None, int(g["d"]), int(g["max_occ"]), T(g["template"]).to(d_),
                                       max_num_nodes=int(g["n"]))
    np.testing.assert_array_equal(out.detach().cpu().numpy(), g["out"])
    Gr = torch.randn(out.shape, generator=torch.Generator().manual_seed(3))
    (out * Gr.to(d_)).sum().backward()
    ent_c = T(g["entity_embeddings"]).requires_grad_(True)
    ref = O.make_start_entity_embeddings(ent_c, T(g["pos"]), int(g["d"]), T(g["template"]), max_num_nodes=int(g["n"]))
    (ref * Gr).sum().backward()
    close(ent.grad, ent_c.grad, atol=1e-5, what="g_entity_embeddings")


def test_start_entity_embeddings_backward_is_a_fixed_order_kernel():
    """the backward of make_start_entity_embeddings (utils/context_utils.py:387-426) runs as kernels — pieces of g * templ,
    then the SpecialSpmmFinal segment walk keyed on the entity ids — no torch op, so two runs are bit-equal (index_add_'s float atomics
    were not) and the result matches the oracle's autograd at RECON's sizes (B = 50, n = 9, d = 8) with heavily repeated entities."""
    from recon_amd.propagation import make_start_entity_embeddings, make_start_embedding
    d_ = dev()
    B, n, d, U = 50, 9, 8, 37
    Cn, S = n * (n - 1), 2 * d * n
    gen = torch.Generator().manual_seed(12)
    ent = torch.randn(U, d, generator=gen)
    pos = torch.randint(0, U, (B, Cn, 2), generator=gen)
    pos[:, :, 0][pos[:, :, 0] % 3 == 0] = 5                            # one entity that owns a third of all first slots: a long segment
    templ = torch.from_numpy(make_start_embedding(n, d)).float().view(1, Cn, S, 1) * (1.0 + torch.rand(1, Cn, S, 1, generator=gen))
    Gr = torch.randn(B, Cn, S, 1, generator=gen)
    grads = []
    for _ in range(2):
        e = ent.to(d_).requires_grad_(True)
        out = make_start_entity_embeddings(e, pos.to(d_), None, d, 0, templ.to(d_), max_num_nodes=n)
        with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CPU]) as prof:
            out.backward(Gr.to(d_))
        grads.append(e.grad.clone())
        ops = {ev.name for ev in prof.events()}
        assert not any("index_add" in o or "index_put" in o or "scatter" in o for o in ops), sorted(ops)
    assert torch.equal(grads[0], grads[1])
    ec = ent.double().requires_grad_(True)
    ref = O.make_start_entity_embeddings(ec, pos, d, templ.double(), max_num_nodes=n)
    (ref * Gr.double()).sum().backward()
    close(grads[0], ec.grad.float(), atol=1e-5, rel_to_max=2e-6, what="g_entity_embeddings (kernel backward)")


@pytest.mark.parametrize("name", ["gcn1_bias", "gcn1_nobias"])
def test_gcn_golden(name):
    """GraphConvolution drop-in (2-D reference form, 72x72 line-graph adjacency) vs the reference."""
    from recon_amd.gcn_layers import GraphConvolution
    g = load_golden(name)
    d_ = dev()
    has_bias = "bias" in g
    layer = GraphConvolution(g["weight"].shape[0], g["weight"].shape[1], bias=has_bias).to(d_)
    sd = {"weight": T(g["weight"])}
    if has_bias:
        sd["bias"] = T(g["bias"])
    layer.load_state_dict(sd, strict=True)
    x = T(g["x"]).to(d_).requires_grad_(True)
    out = layer(x, T(g["adj"]).to(d_))
    close(out, g["out"], atol=1e-5, what="gcn out")
    (out * T(g["G"]).to(d_)).sum().backward()
    close(x.grad, g["g_x"], atol=1e-5, what="g_x")
    close(layer.weight.grad, g["g_weight"], atol=1e-5, what="g_weight")
    if has_bias:
        close(layer.bias.grad, g["g_bias"], atol=1e-5, what="g_bias")


@pytest.mark.parametrize("B,n,I,O_", [(1, 9, 7, 5), (6, 32, 300, 300), (3, 17, 40, 130),
                                      (2, 33, 20, 70), (2, 100, 24, 64), (1, 257, 12, 9)])     # n > 32: tiled adjacency (any n)
def test_gcn_batched_vs_oracle(B, n, I, O_):
    from recon_amd.gcn_layers import GraphConvolution
    d_ = dev()
    g = torch.Generator().manual_seed(B * n)
    x = torch.randn(B, n, I, generator=g)
    adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
    adj = adj / adj.sum(-1, keepdim=True)
    torch.manual_seed(1)
    layer = GraphConvolution(I, O_)
    w, b = layer.weight.detach().clone(), layer.bias.detach().clone()
    Gr = torch.randn(B, n, O_, generator=g)
    xr, adjr, wr, br = (t.clone().requires_grad_(True) for t in (x, adj, w, b))
    ref = O.graph_convolution(xr, adjr, wr, br)
    (ref * Gr).sum().backward()
    layer = layer.to(d_)
    xd, adjd = x.to(d_).requires_grad_(True), adj.to(d_).requires_grad_(True)
    out = layer(xd, adjd)
    close(out, ref, what="gcn batched out")
    (out * Gr.to(d_)).sum().backward()
    close(xd.grad, xr.grad, atol=1e-5, what="g_x")
    close(adjd.grad, adjr.grad, atol=1e-5, what="g_adj")
    close(layer.weight.grad, wr.grad, atol=1e-5, what="g_weight")
    close(layer.bias.grad, br.grad, atol=1e-5, what="g_bias")


@pytest.mark.parametrize("dtype", [torch.float32, torch.bfloat16])
def test_gcn_beyond_4_gib_activations(dtype):
    """Maximum sizes: B = 60 000 graphs x n = 64 nodes x 304 features = 1.17 G elements per activation (4.7 GB in fp32): two copies of
    30 000 graphs must give bit-equal halves (outputs, g_x, g_adj); the first graphs against the oracle; fp32 and bf16 storage."""
    from recon_amd.gcn_layers import GraphConvolution
    d_ = dev()
    if torch.cuda.get_device_properties(0).total_memory < 100 * 2 ** 30:
        pytest.skip("needs ~40 GB of device memory")
    Bh, n, I, O_ = 30000, 64, 304, 304
    g = torch.Generator().manual_seed(9)
    xh = torch.randn(Bh, n, I, generator=g)
    adjh = (torch.rand(Bh, n, n, generator=g) < 0.1).float() + torch.eye(n)
    adjh = adjh / adjh.sum(-1, keepdim=True)
    Grh = torch.randn(Bh, n, O_, generator=g)
    torch.manual_seed(1)
    layer = GraphConvolution(I, O_)
    w, b = layer.weight.detach().clone(), layer.bias.detach().clone()
    layer = layer.to(d_).to(dtype)
    xd = torch.cat([xh, xh]).to(d_).to(dtype).requires_grad_(True)
    adjd = torch.cat([adjh, adjh]).to(d_).to(dtype).requires_grad_(True)
    assert xd.numel() * xd.element_size() > (2 ** 32 if dtype == torch.float32 else 2 ** 31)
    out = layer(xd, adjd)
    (out * torch.cat([Grh, Grh]).to(d_).to(dtype)).sum().backward()
    assert torch.equal(out[:Bh], out[Bh:])
    assert torch.equal(xd.grad[:Bh], xd.grad[Bh:])
    assert torch.equal(adjd.grad[:Bh], adjd.grad[Bh:])
    k = 3
    rnd = (lambda t: t.to(dtype).float())                                # the oracle on the values the layer saw
    xr, adjr, wr, br = (rnd(t).clone().requires_grad_(True) for t in (xh[:k], adjh[:k], w, b))
    ref = O.graph_convolution(xr, adjr, wr, br)
    (ref * rnd(Grh[:k])).sum().backward()
    tol = dict(atol=1e-4, rel_to_max=1e-4) if dtype == torch.float32 else dict(atol=2e-2, rel_to_max=2e-2)
    close(out[:k].float(), ref, what="out", **tol)
    if dtype == torch.float32:       # (bf16: a pre-activation within rounding of zero may sit on the other side of the ReLU, see test_gcn_bf16_*)
        close(xd.grad[:k].float(), xr.grad, what="g_x", **tol)


def test_batches_above_the_grid_limit_run_in_slices(monkeypatch):
    """More than 65 535 graphs per call (the kernels' 16-bit grid dimension) are run in slices by the host side; here the limit is
    lowered to 3 / 2 graphs: outputs and gradients of a 7-graph batch must equal the one-launch results."""
    from recon_amd import propagation, gcn_layers
    from recon_amd.propagation import build_block_adjacency, propagate, make_start_embedding, get_head_indices, get_tail_indices
    d_ = dev()
    n, d, L, B = 4, 2, 2, 7
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(2)
    Ts = [(torch.rand(B, C, dd * dd, generator=g) - 0.4) * 0.5 for _ in range(L)]
    ident0 = torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)
    h00 = torch.randn(B, C, S, 1, generator=g) * torch.from_numpy(make_start_embedding(n, d)).float()
    head = torch.from_numpy(get_head_indices(n, d, bs=B)).to(d_)          # [B, C, 2d] as the reference stores them
    tail = torch.from_numpy(get_tail_indices(n, d, bs=B)).to(d_)
    Gr = torch.randn(B, C, dd * L, generator=g).to(d_)
    x0 = torch.randn(B, 5, 6, generator=g); adj0 = torch.rand(B, 5, 5, generator=g); Gg = torch.randn(B, 5, 4, generator=g).to(d_)
    torch.manual_seed(0)
    layer = gcn_layers.GraphConvolution(6, 4).to(d_)
    res = []
    for lim_p, lim_g in ((65535, 65535), (3, 2)):
        monkeypatch.setattr(propagation, "_MAX_BATCH", lim_p)
        monkeypatch.setattr(gcn_layers, "_MAX_BATCH", lim_g)
        Tl = [t.clone().to(d_).requires_grad_(True) for t in Ts]
        I = ident0.clone().to(d_).requires_grad_(True)
        h = h00.clone().to(d_).requires_grad_(True)
        out = propagate([build_block_adjacency(torch.relu(t), I, n) for t in Tl], h, "tanh", head, tail)
        (out * Gr).sum().backward()
        x = x0.clone().to(d_).requires_grad_(True); adj = adj0.clone().to(d_).requires_grad_(True)
        layer.zero_grad()
        og = layer(x, adj)
        (og * Gg).sum().backward()
        res.append([out, h.grad, I.grad] + [t.grad for t in Tl] + [og, x.grad, adj.grad, layer.weight.grad.clone(), layer.bias.grad.clone()])
    names = ["out", "g_h0", "g_identity", "g_T0", "g_T1", "gcn out", "gcn g_x", "gcn g_adj", "gcn g_weight", "gcn g_bias"]
    for nm, a_, b_ in zip(names, res[0], res[1]):
        if nm in ("g_identity", "gcn g_weight", "gcn g_bias"):            # sums over the batch: slices add in a different order
            close(b_, a_, atol=1e-5, rel_to_max=1e-5, what=nm)
        else:
            assert torch.equal(a_, b_), nm


def test_gcn_inplace_edit_of_result_is_caught():
    """The tensor saved for the backward (its sign is the ReLU mask) is the tensor the caller holds: editing it in place
    must trip autograd's version check instead of silently corrupting the mask."""
    from recon_amd.gcn_layers import GraphConvolution
    d_ = dev()
    layer = GraphConvolution(6, 5).to(d_)
    x = torch.randn(2, 4, 6, device=d_, requires_grad=True)
    adj = torch.rand(2, 4, 4, device=d_)
    out = layer(x, adj)
    assert out.shape == (2, 4, 5)
    out.add_(1.0)
    with pytest.raises(RuntimeError, match="modified by an inplace operation"):
        out.sum().backward()
    x2 = torch.randn(4, 6, device=d_, requires_grad=True)        # the reference's 2-D form keeps its shape too
    assert layer(x2, adj[0]).shape == (4, 5)


def test_propagation_full_size_properties():
    """cfg 3b (B=1024, n=9, 2d=16, L=3): (i) a slice of graphs vs the oracle, (ii) graphs are independent,
    (iii) zero transition matrices + identity diagonal leave h unchanged: relation = head*tail of h0."""
    from recon_amd.propagation import build_block_adjacency, propagate, make_start_embedding, get_head_indices, get_tail_indices
    d_ = dev()
    n, d, L, B = 9, 8, 3, 1024
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(0)
    Ts = [torch.relu(torch.randn(B, C, dd * dd, generator=g)) * 0.1 for _ in range(L)]
    ident = torch.eye(dd)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = torch.randn(B, C, S, 1, generator=g) * tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    adjs = [build_block_adjacency(t.to(d_), ident.to(d_), n) for t in Ts]
    out = propagate(adjs, h0.to(d_), "relu", head.to(d_), tail.to(d_))
    ref = O.propagate([O.build_block_adjacency(t[:3], ident, n) for t in Ts], h0[:3], "relu", head, tail)
    close(out[:3], ref, what="cfg3b slice")
    out2 = propagate([a[5:9].contiguous() for a in adjs], h0[5:9].to(d_), "relu", head.to(d_), tail.to(d_))
    assert torch.equal(out2, out[5:9])
    zero = [build_block_adjacency(torch.zeros(4, C, dd * dd, device=d_), ident.to(d_), n) for _ in range(L)]
    h0p = h0[:4].abs().to(d_)
    o3 = propagate(zero, h0p, "relu", head.to(d_), tail.to(d_))
    flat = h0p.view(4, C, S)
    expect = torch.gather(flat, 2, head.to(d_)[None].expand(4, -1, -1)) * torch.gather(flat, 2, tail.to(d_)[None].expand(4, -1, -1))
    close(o3, expect.repeat(1, 1, L), atol=1e-6, what="identity propagation")


@pytest.mark.parametrize("name", ["gpgnn1_untied", "gpgnn2_tied_n9"])
def test_gpgnn_model_golden(name):
    """the reference GPGNN end to end (stock encoder + HIP block adjacency / propagation): logits and the
    gradient of every trainable parameter against the reference's own outputs."""
    from recon_amd.gpgnn import GPGNN
    g = load_golden(name)
    p = {"max_num_nodes": int(g["n"]), "embedding_dim": int(g["d"]), "layer_number": int(g["L"]), "projection_style": str(g["style"]),
         "non-linear1": "relu", "non-linear": "tanh", "dropout1": 0.0, "position_emb": 3, "units1": 4, "rnn1_layers": 1,
         "bidirectional": 1, "batch_size": int(g["B"])}
    m = GPGNN(p, g["emb"], max_sent_len=4, n_out=3)
    m.load_state_dict({k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")}, strict=False)
    m.train().to(dev())                     # dropout is 0 in the fixture; MIOpen's LSTM backward needs training mode
    out = m(torch.from_numpy(g["sent"]).to(dev()), torch.from_numpy(g["mark"]).to(dev()), None)
    close(out, g["out"], atol=1e-4, what="gpgnn logits")
    (out * torch.from_numpy(g["G"]).to(dev())).sum().backward()
    for k, v in m.named_parameters():
        if "g." + k in g:
            close(v.grad, g["g." + k], atol=1e-4, rel_to_max=1e-4, what="grad " + k)


def test_recon_eac_model_golden():
    """SURVEY 8f N3, second model: the reference RECON_EAC end to end — entity-context encoder (stock ops), per-batch start
    embeddings, block adjacency and propagation (HIP) — logits and every parameter gradient, including those that reach the
    context encoder only through the start-embedding kernel's backward."""
    from recon_amd.gpgnn import RECON_EAC
    from tests.test_host_cpu import EAC_P
    g = load_golden("eac1_untied")
    m = RECON_EAC(dict(EAC_P), g["emb"], max_sent_len=4, n_out=3, char_vocab=list(range(int(g["n_chars"]))))
    m.load_state_dict({k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")}, strict=False)
    m.train().to(dev())
    t = lambda k: torch.from_numpy(g[k]).to(dev())
    out = m(t("sent"), t("mark"), None, None, None, t("ctx_words"), t("ctx_chars"), t("ctx_mask"), t("pos"), int(g["max_occ"]))
    close(out, g["out"], atol=1e-4, what="recon_eac logits")
    (out * t("G")).sum().backward()
    seen = 0
    for k, v in m.named_parameters():
        if "g." + k in g:
            close(v.grad, g["g." + k], atol=1e-4, rel_to_max=1e-4, what="grad " + k)
            seen += 1
    assert seen == sum(k.startswith("g.") for k in g)


def _shell_inputs(g):
    t = lambda k: torch.from_numpy(g[k]).to(dev())
    return t, (t("sent"), t("mark"), None, None, None, t("ctx_words"), t("ctx_chars"), t("ctx_mask"), t("pos"), int(g["max_occ"]))


def _check_model_grads(m, g, out, what):
    close(out, g["out"], atol=1e-4, what=what + " logits")
    (out * torch.from_numpy(g["G"]).to(out.device)).sum().backward()
    seen = 0
    for k, v in m.named_parameters():
        if "g." + k in g:
            close(v.grad, g["g." + k], atol=1e-4, rel_to_max=1e-4, what=what + " grad " + k)
            seen += 1
    assert seen == sum(k.startswith("g.") for k in g)


def test_recon_eac_kggat_model_golden():
    """Wider N3: the reference RECON_EAC_KGGAT (models/models.py:489-701) end to end: propagation features + the KB-GAT entity
    embeddings of every pair in front of the classifier; logits and every parameter gradient; the tied branch fails as it does there."""
    from recon_amd.gpgnn import RECON_EAC_KGGAT
    from tests.test_host_cpu import KGGAT_P
    g = load_golden("kggat1_untied")
    m = RECON_EAC_KGGAT(dict(KGGAT_P), g["emb"], max_sent_len=4, n_out=3, char_vocab=list(range(int(g["n_chars"]))))
    missing, unexpected = m.load_state_dict({k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")}, strict=False)
    assert set(missing) == {"head_indices", "tail_indices", "start_embedding"} and not unexpected
    m.train().to(dev())
    t, args = _shell_inputs(g)
    _check_model_grads(m, g, m(*args, t("gat")), "recon_eac_kggat")
    tied = RECON_EAC_KGGAT(dict(KGGAT_P, projection_style="tie"), g["emb"], max_sent_len=4, n_out=3, char_vocab=list(range(int(g["n_chars"])))).to(dev())
    with pytest.raises(RuntimeError):
        tied(*args, t("gat"))


def test_recon_full_model_golden():
    """Wider N3: the reference's full model RECON (models/models.py:703-968): + the L1 translation residuals in every output
    relation's space, scattered to the pairs with known embeddings.  `gat_relation_embeddings` trains, `W_ent2rel` does not."""
    from recon_amd.gpgnn import RECON
    from tests.test_host_cpu import KGGAT_P, recon_constructor_tables
    g = load_golden("recon1_untied")
    m = RECON(dict(KGGAT_P), g["emb"], 4, 3, list(range(int(g["n_chars"]))), *recon_constructor_tables(g))
    sd = {k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")}
    np.testing.assert_array_equal(m.gat_relation_embeddings.detach().numpy(), sd["gat_relation_embeddings"].numpy())     # the constructor's own table lookup
    np.testing.assert_array_equal(m.W_ent2rel.numpy(), sd["W_ent2rel"].numpy())
    missing, unexpected = m.load_state_dict(sd, strict=False)
    assert set(missing) == {"start_embedding"} and not unexpected            # head / tail indices are not checkpoint keys in this class (:759-768)
    m.train().to(dev())
    assert m.head_indices.is_cuda
    t, args = _shell_inputs(g)
    out = m(*args, t("nz"), t("nz_pos"), t("gat"))
    assert out.shape == (4 * 6, 3)
    _check_model_grads(m, g, out, "recon")
    assert m.W_ent2rel.grad is None
    # no pair with known embeddings: the score block stays zero
    out0 = m(*args, t("nz")[:0], t("nz_pos")[:0], t("gat"))
    assert torch.isfinite(out0).all()


# ------------------------------------------------------------------------------- GraphConvolution in bfloat16 (configs[2])
def _bf(t):
    return t.to(torch.bfloat16)


@pytest.mark.parametrize("adj_grad", [True, False])        # False: the forward is the fused kernel (n <= 32, out <= 320: `support` never leaves the registers)
@pytest.mark.parametrize("B,n,I,O_", [(1, 9, 7, 5), (6, 32, 300, 300), (3, 17, 40, 136), (2, 100, 24, 64), (4, 32, 304, 208), (5, 31, 33, 320), (7, 2, 8, 16)])
def test_gcn_bf16_vs_oracle(B, n, I, O_, adj_grad):
    """bf16 storage / fp32 accumulation.  Forward: against the fp32 oracle on the SAME bf16-rounded inputs (what remains is the
    rounding of `support` and of the result to bf16, 2^-8 relative each).  Backward: against the oracle's gradient formulas
    evaluated with the ReLU mask of the bf16 forward — a pre-activation within bf16 rounding of zero may land on the other side
    of the ReLU than in fp32, which changes that element's gradient by O(1) and says nothing about the kernels."""
    from recon_amd.gcn_layers import GraphConvolution
    d_ = dev()
    g = torch.Generator().manual_seed(B * n + I)
    x = _bf(torch.randn(B, n, I, generator=g))
    adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
    adj = _bf(adj / adj.sum(-1, keepdim=True))
    torch.manual_seed(1)
    layer = GraphConvolution(I, O_).to(torch.bfloat16)
    w, b = layer.weight.detach().clone().float(), layer.bias.detach().clone().float()
    Gr = _bf(torch.randn(B, n, O_, generator=g))
    ref = O.graph_convolution(x.float(), adj.float(), w, b)
    layer = layer.to(d_)
    xd, adjd = x.to(d_).requires_grad_(True), adj.to(d_).requires_grad_(adj_grad)
    out = layer(xd, adjd)
    assert out.dtype == torch.bfloat16 and out.shape == (B, n, O_)
    close(out.float(), ref, atol=1e-3, rel_to_max=1.5e-2, what="gcn bf16 out")
    if not adj_grad:                                                   # the fused and the two-kernel forward round the same quantities to bf16
        out2 = layer(x.to(d_), adj.to(d_).requires_grad_(True))
        close(out.float(), out2.float(), atol=1e-3, rel_to_max=1e-2, what="fused vs unfused forward")
    flipped = ((out.float().cpu() > 0) != (ref > 0)).float().mean().item()
    assert flipped < 0.02, flipped                                     # the masks agree except next to zero
    (out * Gr.to(d_)).sum().backward()
    # models/layers.py:57-63 differentiated by hand (as oracle.graph_convolution's autograd does), mask from the bf16 forward
    sup = (x.float() @ w).to(torch.bfloat16).float()                   # torch.mm in bf16 rounds its result: so does the kernel
    gpre = Gr.float() * (out.float().cpu() > 0)
    g_sup = (adj.float().transpose(1, 2) @ gpre).to(torch.bfloat16).float()
    close(xd.grad.float(), g_sup @ w.t(), atol=1e-3, rel_to_max=1e-2, what="g_x")
    if adj_grad:
        close(adjd.grad.float(), gpre @ sup.transpose(1, 2), atol=1e-3, rel_to_max=1e-2, what="g_adj")
    close(layer.weight.grad.float(), x.float().reshape(-1, I).t() @ g_sup.reshape(-1, O_), atol=1e-3, rel_to_max=1e-2, what="g_weight")
    close(layer.bias.grad.float(), gpre.reshape(-1, O_).sum(0), atol=1e-3, rel_to_max=1e-2, what="g_bias")


def test_gcn_bf16_stack_reads_padded_rows_in_place():
    """Three bf16 layers at D = 300 (cfg 3a): a layer's result is a [..., :300] view of 304-wide rows that the next layer (and, on
    the way back, the previous one) reads in place; the 2-D reference form and a repacked (contiguous) input agree with it."""
    from recon_amd.gcn_layers import GraphConvolution, _rows_view
    d_ = dev()
    B, n, D = 8, 32, 300
    g = torch.Generator().manual_seed(5)
    x = _bf(torch.randn(B, n, D, generator=g)).to(d_).requires_grad_(True)
    adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
    adj = _bf(adj / adj.sum(-1, keepdim=True)).to(d_)
    torch.manual_seed(2)
    layers = [GraphConvolution(D, D).to(torch.bfloat16).to(d_) for _ in range(3)]
    h = x
    for layer in layers:
        h = layer(h, adj)
        assert _rows_view(h, D) == 304 and not h.is_contiguous()
    G = _bf(torch.randn(B, n, D, generator=g)).to(d_)
    (h * G).sum().backward()
    # the same stack with every intermediate repacked to a contiguous tensor
    x2 = x.detach().clone().requires_grad_(True)
    h2 = x2
    for layer in layers:
        h2 = layer(h2, adj).contiguous()
    grads = [layer.weight.grad.clone() for layer in layers]
    for layer in layers:
        layer.weight.grad = None
    (h2 * G).sum().backward()
    assert torch.equal(h, h2) and torch.equal(x.grad, x2.grad)
    for layer, gw in zip(layers, grads):
        assert torch.equal(layer.weight.grad, gw)
    # fp32 oracle on the bf16-rounded parameters
    hr = x.detach().cpu().float()
    for layer in layers:
        hr = O.graph_convolution(hr, adj.cpu().float(), layer.weight.detach().cpu().float(), layer.bias.detach().cpu().float())
    close(h.float(), hr, atol=1e-3, rel_to_max=3e-2, what="3-layer bf16 stack")
    y2d = layers[0](x.detach()[0], adj[0])
    assert torch.equal(y2d, layers[0](x.detach(), adj)[0])


def test_gcn_stack_takes_a_ragged_batch():
    """gcn_stack() with a RaggedAdjacency (BASELINE.json configs[4]'s graphs of different sizes): the layer loop, same bits as calling the layers."""
    from recon_amd.gcn_layers import GraphConvolution, RaggedAdjacency, gcn_stack
    d_ = dev()
    g = torch.Generator().manual_seed(2)
    sizes = [5, 40, 17, 256, 33]
    mats = [_bf(torch.rand(n, n, generator=g) / n).to(d_) for n in sizes]
    rag = RaggedAdjacency.from_dense(mats)
    x = _bf(torch.randn(sum(sizes), 48, generator=g)).to(d_).requires_grad_(True)
    torch.manual_seed(8)
    layers = [GraphConvolution(48 if l == 0 else 64, 64).to(torch.bfloat16).to(d_) for l in range(3)]
    y = gcn_stack(x, rag, layers)
    h = x
    for l in layers:
        h = l(h, rag)
    assert torch.equal(y, h)
    y.float().sum().backward()
    assert x.grad is not None and torch.isfinite(x.grad.float()).all()


def test_gcn_stack_bf16_trains_with_a_frozen_layer_and_a_padded_input():
    """The training form of gcn_stack() with (i) a frozen layer in the middle — its weight-gradient product drops out of the one split-K
    launch, the others keep their places in the partial workspace — and (ii) an input that is itself the padded output of a bf16
    GraphConvolution (row stride 304 for 300 features: read in place, no aligned copy written).  Everything that has a gradient carries the
    bits (g_W: the values) of the layer loop."""
    import recon_amd.gcn_layers as GL
    from recon_amd.gcn_layers import GraphConvolution, gcn_stack
    d_ = dev()
    B, n, D, L = 6, 32, 300, 3
    g = torch.Generator().manual_seed(11)
    x0 = _bf(torch.randn(B, n, 64, generator=g)).to(d_)
    adj = (torch.rand(B, n, n, generator=g) < 0.2).float() + torch.eye(n)
    adj = _bf(adj / adj.sum(-1, keepdim=True)).to(d_)
    torch.manual_seed(6)
    pre = GraphConvolution(64, D).to(torch.bfloat16).to(d_)
    layers = [GraphConvolution(D, D).to(torch.bfloat16).to(d_) for _ in range(L)]
    layers[1].weight.requires_grad_(False)
    layers[1].bias.requires_grad_(False)
    G = _bf(torch.randn(B, n, D, generator=g)).to(d_)
    params = [p for m in [pre] + layers for p in m.parameters()]

    def run(fn):
        for p in params:
            p.grad = None
        h = pre(x0, adj)                                            # [B, n, 300] view of rows of 304: `_recon_padded`
        assert h.stride(-2) == 304
        y = fn(h)
        y.backward(G)
        return [y.detach().clone()] + [p.grad.clone() if p.grad is not None else None for p in params]
    a = run(lambda h: gcn_stack(h, adj, layers))

    def loop(h):
        for l in layers:
            h = l(h, adj)
        return h
    b = run(loop)
    assert a[3 + 2] is None and a[3 + 3] is None and b[3 + 2] is None          # the frozen layer's parameters (after pre's two and layer 0's two)
    for i, (u, v) in enumerate(zip(a, b)):
        if u is None:
            continue
        if u.dim() == 2 and i >= 3:                                    # stack weights: another split-K order
            close(u.float(), v.float(), atol=1e-4, rel_to_max=1.6e-2, what="tensor %d (a weight gradient of the stack)" % i)
        else:
            assert torch.equal(u, v), "tensor %d differs from the loop: max %g" % (i, (u.float() - v.float()).abs().max().item())


@pytest.mark.parametrize("B,n,I,D,L", [(9, 32, 300, 300, 3), (5, 32, 40, 136, 2), (7, 8, 300, 64, 4), (1030, 32, 300, 300, 3), (3, 12, 22, 320, 3), (6, 31, 300, 300, 3)])
def test_gcn_stack_bf16_is_bit_equal_to_the_layers(B, n, I, D, L, monkeypatch):
    """gcn_stack(): L GraphConvolutions over one adjacency in one launch, the activations in LDS between the layers — the bits of the
    layer-by-layer loop (same fragments, same MFMA order), and within bf16 rounding of the fp32 oracle.  (n = 31: not a multiple of 4,
    the loop runs.)  Parts = 2 and 4 waves per graph."""
    from recon_amd.gcn_layers import GraphConvolution, gcn_stack
    d_ = dev()
    g = torch.Generator().manual_seed(B + n + I)
    x = _bf(torch.randn(B, n, I, generator=g)).to(d_)
    adj = (torch.rand(B, n, n, generator=g) < 0.2).float() + torch.eye(n)
    adj = _bf(adj / adj.sum(-1, keepdim=True)).to(d_)
    torch.manual_seed(4)
    layers = [GraphConvolution(I if l == 0 else D, D).to(torch.bfloat16).to(d_).eval() for l in range(L)]
    with torch.no_grad():
        fused = gcn_stack(x, adj, layers)
        h = x
        for l in layers:
            h = l(h, adj)
    assert fused.shape == h.shape and fused.dtype == torch.bfloat16
    assert torch.equal(fused, h), "fused stack differs from the layer loop: max %g" % (fused.float() - h.float()).abs().max().item()
    hr = x.float().cpu()
    for l in layers:
        hr = O.graph_convolution(hr, adj.float().cpu(), l.weight.detach().float().cpu(), l.bias.detach().float().cpu())
    close(fused.float(), hr, atol=2e-3, rel_to_max=4e-2, what="stack vs oracle")
    y = gcn_stack(x.requires_grad_(True), adj, layers)                   # gradients wanted: the training form (or the loop), with autograd
    y.float().sum().backward()
    assert x.grad is not None


@pytest.mark.parametrize("B,n,I,D,L,bias,xgrad", [(9, 32, 300, 300, 3, True, True), (5, 32, 40, 136, 2, True, False), (7, 8, 300, 64, 4, False, True),
                                                   (1030, 32, 300, 300, 3, True, True), (3, 12, 22, 320, 3, True, True), (6, 28, 37, 300, 2, True, True)])
def test_gcn_stack_bf16_trains_bit_equal_to_the_layers(B, n, I, D, L, bias, xgrad, monkeypatch):
    """gcn_stack() under autograd: one launch forward (every layer's result kept), one launch backward + one split-K launch for all weight
    gradients (models/layers.py:57-63 applied L times + autograd).  Output, g_x and g_bias carry the bits of the layer-by-layer loop, g_W
    its values up to the summation order of the split-K partials; all lie within bf16 rounding of the fp32 oracle's autograd, and a second
    run gives the same bits.  Odd in_features (37: rows repacked), a non-contiguous output gradient, layers
    without bias, an input that needs no gradient."""
    import recon_amd.gcn_layers as GL
    from recon_amd.gcn_layers import GraphConvolution, gcn_stack
    d_ = dev()
    g = torch.Generator().manual_seed(B + n + I + 1)
    x0 = _bf(torch.randn(B, n, I, generator=g)).to(d_)
    adj = (torch.rand(B, n, n, generator=g) < 0.2).float() + torch.eye(n)
    adj = _bf(adj / adj.sum(-1, keepdim=True)).to(d_)
    torch.manual_seed(5)
    layers = [GraphConvolution(I if l == 0 else D, D, bias=bias).to(torch.bfloat16).to(d_).train() for l in range(L)]
    G = _bf(torch.randn(B, n, 2 * D, generator=g)).to(d_)[..., ::2]        # a strided gradient: repacked inside

    def run(fn):
        for l in layers:
            l.weight.grad = None
            if l.bias is not None:
                l.bias.grad = None
        x = x0.clone().requires_grad_(xgrad)
        y = fn(x)
        y.backward(G)
        return (y.detach().clone(), x.grad.clone() if xgrad else None, [l.weight.grad.clone() for l in layers],
                [l.bias.grad.clone() if l.bias is not None else None for l in layers])

    calls = []
    real = GL._GcnB16StackFunction.apply
    monkeypatch.setattr(GL._GcnB16StackFunction, "apply", staticmethod(lambda *a: (calls.append(1), real(*a))[1]))
    yf, gxf, gwf, gbf = run(lambda x: gcn_stack(x, adj, layers))
    assert calls, "the training form of gcn_stack() was not taken"

    y2, gx2, gw2, gb2 = run(lambda x: gcn_stack(x, adj, layers))
    assert torch.equal(yf, y2) and all(torch.equal(a, b) for a, b in zip(gwf, gw2)), "the stack is not run-to-run reproducible"

    def loop(x):
        for l in layers:
            x = l(x, adj)
        return x
    yl, gxl, gwl, gbl = run(loop)
    assert torch.equal(yf, yl), "output differs from the loop: max %g" % (yf.float() - yl.float()).abs().max().item()
    if xgrad:
        assert torch.equal(gxf, gxl), "g_x differs from the loop: max %g" % (gxf.float() - gxl.float()).abs().max().item()
    for l in range(L):
        # the stack's weight gradients run as ONE split-K launch with its own split count: same products, another (fixed) summation order
        close(gwf[l].float(), gwl[l].float(), atol=1e-4, rel_to_max=1.6e-2, what="g_W[%d] vs the loop" % l)
        if bias:
            assert torch.equal(gbf[l], gbl[l]), "g_bias[%d] differs from the loop" % l
    # models/layers.py:57-63 differentiated by hand, layer by layer (as oracle.graph_convolution's autograd does), on the bf16-rounded
    # parameters and with the ReLU masks of the bf16 forward (see test_gcn_bf16_vs_oracle: a pre-activation within bf16 rounding of zero
    # may land on the other side of the ReLU than in fp32, which says nothing about the kernels); the forward itself against the oracle
    adjc = adj.float().cpu()
    with torch.no_grad():
        acts, h = [], x0
        for l in layers:
            h = l(h, adj)
            acts.append(h.float().cpu())
    hr = x0.float().cpu()
    for l in layers:
        hr = O.graph_convolution(hr, adjc, l.weight.detach().float().cpu(), l.bias.detach().float().cpu() if bias else None)
    close(yf.float(), hr, atol=2e-3, rel_to_max=1.5e-2 * L, what="stack output vs oracle")
    gcur = G.float().cpu()
    for l in range(L - 1, -1, -1):
        xin = (acts[l - 1] if l > 0 else x0.float().cpu())
        W = layers[l].weight.detach().float().cpu()
        gpre = gcur * (acts[l] > 0)
        g_sup = (adjc.transpose(1, 2) @ gpre).to(torch.bfloat16).float()
        close(gwf[l].float(), xin.reshape(-1, W.shape[0]).t() @ g_sup.reshape(-1, D), atol=2e-3, rel_to_max=1.5e-2, what="stack g_W[%d] vs oracle" % l)
        if bias:
            close(gbf[l].float(), gpre.reshape(-1, D).sum(0), atol=2e-3, rel_to_max=1.5e-2, what="stack g_bias[%d] vs oracle" % l)
        gcur = (g_sup @ W.t()).to(torch.bfloat16).float()
    if xgrad:
        close(gxf.float(), gcur, atol=2e-3, rel_to_max=1e-2 * L, what="stack g_x vs oracle")


@pytest.mark.parametrize("sizes,I,O_", [([5, 32, 17, 256, 100, 1, 33], 40, 136), ([256, 256, 200], 300, 300), ([1], 7, 5), ([3] * 70, 24, 16), ([129, 64], 33, 64)])
def test_gcn_bf16_ragged_vs_oracle(sizes, I, O_):
    """Graphs of different sizes in one call (BASELINE.json configs[4]: up to 256 nodes per graph): x @ W over all node rows, the
    aggregate per graph with its own dense adjacency.  Forward against the fp32 oracle per graph on the bf16 operands; gradients from
    models/layers.py:57-63 differentiated by hand under the forward's own ReLU mask (the test_gcn_bf16_vs_oracle method)."""
    from recon_amd.gcn_layers import GraphConvolution, RaggedAdjacency
    d_ = dev()
    g = torch.Generator().manual_seed(sum(sizes) + I)
    N = sum(sizes)
    x = _bf(torch.randn(N, I, generator=g))
    mats = []
    for n in sizes:
        a = (torch.rand(n, n, generator=g) < max(0.05, 4.0 / n)).float() + torch.eye(n)
        mats.append(_bf(a / a.sum(-1, keepdim=True)))
    torch.manual_seed(1)
    layer = GraphConvolution(I, O_).to(torch.bfloat16)
    w, b = layer.weight.detach().clone().float(), layer.bias.detach().clone().float()
    Gr = _bf(torch.randn(N, O_, generator=g))
    layer = layer.to(d_)
    xd = x.to(d_).requires_grad_(True)
    rag = RaggedAdjacency.from_dense([m.to(d_) for m in mats])
    rag.values.requires_grad_(True)
    out = layer(xd, rag)
    assert out.dtype == torch.bfloat16 and out.shape == (N, O_)
    (out * Gr.to(d_)).sum().backward()
    outc, gx, gv = out.detach().float().cpu(), xd.grad.float().cpu(), rag.values.grad.float().cpu()
    gw_ref, gb_ref = torch.zeros(I, O_), torch.zeros(O_)
    r0, a0 = 0, 0
    for n, m in zip(sizes, mats):
        xs, mf = x[r0:r0 + n].float(), m.float()
        ref = O.graph_convolution(xs, mf, w, b)
        close(outc[r0:r0 + n], ref, atol=1e-3, rel_to_max=1.5e-2, what="ragged out (n=%d)" % n)
        sup = (xs @ w).to(torch.bfloat16).float()
        gpre = Gr[r0:r0 + n].float() * (outc[r0:r0 + n] > 0)
        g_sup = (mf.t() @ gpre).to(torch.bfloat16).float()
        close(gx[r0:r0 + n], g_sup @ w.t(), atol=1e-3, rel_to_max=1e-2, what="ragged g_x (n=%d)" % n)
        close(gv[a0:a0 + n * n].view(n, n), gpre @ sup.t(), atol=1e-3, rel_to_max=1e-2, what="ragged g_adj (n=%d)" % n)
        gw_ref += xs.t() @ g_sup
        gb_ref += gpre.sum(0)
        r0, a0 = r0 + n, a0 + n * n
    close(layer.weight.grad.float(), gw_ref, atol=1e-3, rel_to_max=1e-2, what="ragged g_weight")
    close(layer.bias.grad.float(), gb_ref, atol=1e-3, rel_to_max=1e-2, what="ragged g_bias")
    if len(set(sizes)) == 1:                                             # equal sizes: the batched [B, n, in] call computes the same thing
        n = sizes[0]
        with torch.no_grad():
            dense = layer(x.to(d_).view(len(sizes), n, I), torch.stack([m.to(d_) for m in mats]).requires_grad_(True))
        close(dense.reshape(N, O_).float(), outc, atol=1e-3, rel_to_max=1e-2, what="ragged vs batched")


def test_gcn_bf16_foreign_padded_view_is_repacked():
    """A caller's own `buf[..., :300]` view of wider rows may hold anything in its pad columns (here: NaN); the bf16 kernels read pad
    columns in place only for rows this module produced, everything else is repacked with zeros — the result must not depend on them."""
    from recon_amd.gcn_layers import GraphConvolution
    d_ = dev()
    B, n, D = 4, 32, 300
    g = torch.Generator().manual_seed(11)
    buf = torch.full((B, n, 304), float("nan"), dtype=torch.bfloat16, device=d_)
    x = _bf(torch.randn(B, n, D, generator=g)).to(d_)
    buf[..., :D] = x
    adj = (torch.rand(B, n, n, generator=g) < 0.15).float() + torch.eye(n)
    adj = _bf(adj / adj.sum(-1, keepdim=True)).to(d_)
    torch.manual_seed(2)
    layer = GraphConvolution(D, D).to(torch.bfloat16).to(d_)
    y_view, y_plain = layer(buf[..., :D], adj), layer(x, adj)
    assert torch.isfinite(y_view.float()).all() and torch.equal(y_view, y_plain)
    # inference: the fused kernel reads both in place (the view's NaN pads and the unpadded rows' neighbours sit behind the K-tail mask)
    with torch.no_grad():
        z_view, z_plain = layer(buf[..., :D], adj), layer(x, adj)
        big = x.clone()
        big[:, 1:, :20] = float("inf")                                   # what an unpadded row's tail fragment would touch in the NEXT row
        z_row0 = layer(big, adj * torch.eye(n, device=d_, dtype=torch.bfloat16))      # identity-pattern adj: row 0 of each graph depends on its own x only
    assert torch.isfinite(z_view.float()).all() and torch.equal(z_view, z_plain)
    close(z_plain.float(), y_plain.float(), atol=1e-3, rel_to_max=1e-2, what="inference vs training forward")
    assert torch.isfinite(z_row0[:, 0].float()).all()


# ------------------------------------------------------------------------------- wide-state float32 backward on the two-term f16 kernels
@pytest.mark.parametrize("n,L,B,act,per_batch", [
    (12, 3, 3, "relu", True),        # S = 192, C = 132: RT = 2, three channel chunks (the last of 4 channels), d A in one pass of 8 K steps
    (11, 2, 9, "tanh", False),       # S = 176: partial last row tile; shared h0; more than 8 graphs
    (17, 3, 2, "relu", True),        # S = 272, C = 272: RT = 3, odd K steps padded; d A in one pass of 16 K steps (10 used)
    (24, 2, 2, "linear", True),      # S = 384, C = 552: d A in two passes
    (13, 1, 300, "relu", False),     # one hop; two slices of the split workspace
])
def test_wide_backward_chain_form_vs_oracle(n, L, B, act, per_batch):
    """160 < S <= 512 with GP-GNN's block-structured gather indices (utils/embedding_utils.py:184-202): the backward's chain
    d loss / d H^l-1 = A_l^T Y_l runs on the forward's two-term f16 kernel over transposed split adjacencies, the d A_l products on its mirror
    image (csrc/prop_hl.hip: k_propagate_fwd_hl<.., true>, k_prop_gadj_hl).  Arbitrary dense adjacencies with rows and columns of very
    different magnitude; every gradient against the float64 oracle."""
    import ctypes as C
    from recon_amd import _lib
    from recon_amd.propagation import propagate, get_head_indices, get_tail_indices
    d_ = dev()
    d = 8
    Cn, S, dd = n * (n - 1), 16 * n, 16
    g = torch.Generator().manual_seed(100 * n + L)
    adjs = [(torch.rand(B, S, S, generator=g) - 0.45) * (2.0 / S ** 0.5) * (1 + l) for l in range(L)]
    for a in adjs:
        a[:, :, ::7] *= 8.0
        a[:, 5] *= 1e-3
    h0 = torch.randn(B, Cn, S, 1, generator=g) if per_batch else torch.randn(Cn, S, 1, generator=g)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Gr = torch.randn(B, Cn, dd * L, generator=g)
    Gr[:, 3] = 0.0                                                     # a channel without gradient: its Y rows are zero (scale of a zero row)

    probe = _lib.PropArgs(B, Cn, S, L, dd, _lib.ACT[act], None, None, 0, None, None, 0, None, None, None, None, None, None, 0)
    probe.split_ws_bytes = _lib.lib().recon_propagate_ws_bytes(C.byref(probe))
    probe.split_ws = 1                                                 # any non-null value: the query looks at the size
    assert _lib.lib().recon_propagate_bwd_chain_ws_floats(C.byref(probe)) > 0, "chain form not offered for this shape"

    def run(device, prop, dt):
        A = [a.clone().to(device=device, dtype=dt).requires_grad_(True) for a in adjs]
        h = h0.clone().to(device=device, dtype=dt).requires_grad_(True)
        out = prop(A, h, act, head.to(device), tail.to(device))
        (out * Gr.to(device=device, dtype=dt)).sum().backward()
        return out.detach(), [a.grad for a in A], h.grad
    out_r, gA_r, gh_r = run("cpu", lambda *a: O.propagate(*a, as_gemm=True), torch.float64)
    out_h, gA_h, gh_h = run(d_, propagate, torch.float32)
    close(out_h, out_r.float(), atol=1e-4, rel_to_max=1e-5, what="out")
    for l in range(L):
        close(gA_h[l], gA_r[l].float(), atol=1e-5, what="chain g_adj[%d]" % l)
    close(gh_h, gh_r.float(), atol=1e-5, what="chain g_h0")


@pytest.mark.parametrize("n,L,B,act,per_batch", [
    (12, 3, 3, "relu", True),        # S = 192: RT = 2
    (17, 2, 2, "tanh", False),       # S = 272: RT = 3; shared h0
    (24, 2, 2, "relu", True),        # S = 384, C = 552: d T in two K passes
    (11, 1, 300, "tanh", False),     # one hop; two slices of the split workspace (tanh: among 6 M pre-activations one lands within the two-term
                                     # arithmetic's 1e-7 of zero and flips its ReLU mask against the float64 oracle — seen with this seed)
])
def test_propagate_blocks_trains_at_wide_states(n, L, B, act, per_batch):
    """propagate_blocks() with gradients for 10 < n <= 32 in float32 (csrc/prop_hl.hip in block mode, both directions): against the float64
    oracle, and d identity / d T against the route through the materialised adjacency (RECON_PROP_BLOCKS=0 is process-wide: compared through
    propagate + build_block_adjacency here)."""
    from recon_amd.propagation import propagate_blocks, propagate, build_block_adjacency, get_head_indices, get_tail_indices, make_start_embedding, _blocks_wide_trainable
    d_ = dev()
    d = 8
    Cn, S, dd = n * (n - 1), 16 * n, 16
    g = torch.Generator().manual_seed(7 * n + L)
    Ts = [torch.relu(torch.randn(B, Cn, dd * dd, generator=g)) * (0.6 / n) for _ in range(L)]
    for t in Ts:
        t[:, ::5] *= 6.0                                               # blocks of very different magnitude
    ident = torch.eye(dd) + 0.02 * torch.randn(dd, dd, generator=g)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = (torch.randn(B, Cn, S, 1, generator=g) if per_batch else torch.randn(Cn, S, 1, generator=g)) * tmpl
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Gr = torch.randn(B, Cn, dd * L, generator=g)
    assert _blocks_wide_trainable(B, n, dd, h0.to(d_), L, head.to(d_), tail.to(d_))

    def run(device, dt, fn):
        Tl = [t.clone().to(device=device, dtype=dt).requires_grad_(True) for t in Ts]
        I = ident.clone().to(device=device, dtype=dt).requires_grad_(True)
        h = h0.clone().to(device=device, dtype=dt).requires_grad_(True)
        out = fn(Tl, I, h, head.to(device), tail.to(device))
        (out * Gr.to(device=device, dtype=dt)).sum().backward()
        return out.detach(), [t.grad for t in Tl], I.grad, h.grad
    ref = run("cpu", torch.float64, lambda Tl, I, h, hd, tl: O.propagate([O.build_block_adjacency(t, I, n) for t in Tl], h, act, hd, tl, as_gemm=True))
    blk = run(d_, torch.float32, lambda Tl, I, h, hd, tl: propagate_blocks(Tl, I, n, h, act, hd, tl))
    dense = run(d_, torch.float32, lambda Tl, I, h, hd, tl: propagate([build_block_adjacency(t, I, n) for t in Tl], h, act, hd, tl))
    for name, got in (("blocks", blk), ("dense", dense)):
        close(got[0], ref[0].float(), atol=1e-4, rel_to_max=1e-5, what=name + " out")
        for l in range(L):
            close(got[1][l], ref[1][l].float(), atol=1e-5, what="%s g_T[%d]" % (name, l))
        close(got[2], ref[2].float(), atol=1e-5, rel_to_max=2e-5, what=name + " g_identity")
        close(got[3], ref[3].float(), atol=1e-5, what=name + " g_h0")


def test_wide_backward_takes_a_gradient_view_at_an_odd_offset():
    """The wide-state backward reads grad_out in 16-byte pieces: a contiguous view that starts 4 bytes into its storage is copied first — same
    gradients, bit for bit, as with an aligned tensor."""
    from recon_amd.propagation import propagate, get_head_indices, get_tail_indices
    d_ = dev()
    n, L, B = 12, 2, 2
    Cn, S, dd = n * (n - 1), 16 * n, 16
    g = torch.Generator().manual_seed(5)
    adjs = [((torch.rand(B, S, S, generator=g) - 0.45) * (2.0 / S ** 0.5)).to(d_) for _ in range(L)]
    h0 = torch.randn(B, Cn, S, 1, generator=g).to(d_)
    head = torch.from_numpy(get_head_indices(n, 8, bs=1)[0]).to(d_)
    tail = torch.from_numpy(get_tail_indices(n, 8, bs=1)[0]).to(d_)
    Gr = torch.randn(B * Cn * dd * L + 1, generator=g).to(d_)
    odd = Gr[1:].view(B, Cn, dd * L)
    assert odd.data_ptr() % 16 != 0 and odd.is_contiguous()
    res = []
    for G in (odd, odd.clone()):
        A = [a.clone().requires_grad_(True) for a in adjs]
        h = h0.clone().requires_grad_(True)
        propagate(A, h, "relu", head, tail).backward(G)
        res.append([a.grad for a in A] + [h.grad])
    for x, y in zip(*res):
        assert torch.equal(x, y)
