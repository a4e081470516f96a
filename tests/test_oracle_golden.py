"""Pins oracle/recon_oracle.py against the golden vectors produced by running the reference
(tests/golden/gen_golden.py).  CPU only."""
import numpy as np
import pytest
import torch

from conftest import load_golden, hashed_uniform
from oracle import recon_oracle as O

T = torch.from_numpy


def _tol(dtype):
    return dict(atol=1e-12, rtol=1e-10) if dtype == np.float64 else dict(atol=2e-5, rtol=1e-5)


GAT_CASES = ["gat1_cfg1", "gat2_dups", "gat3_nhop", "gat4_noconcat", "gat5_train", "gat6_dups_f64",
             "gat7_cfg2_slice"]


@pytest.mark.parametrize("name", GAT_CASES)
@pytest.mark.parametrize("aten", [False, True])
def test_gat_layer_forward_backward(name, aten):
    g = load_golden(name)
    x, ee, a, a2 = (T(g[k]).requires_grad_(True) for k in ("x", "edge_embed", "a", "a_2"))
    edge = T(g["edge"])
    nhop = T(g["edge_nhop"]) if "edge_nhop" in g else None
    ee2 = T(g["edge_embed_nhop"]).requires_grad_(True) if nhop is not None else None
    mask = T(g["mask"]) if float(g["train_p"]) > 0 else None
    out = O.gat_layer_forward(x, edge, ee, nhop, ee2, a, a2, float(g["alpha"]), bool(g["concat"]),
                              mask=mask, aten_sequence=aten)
    tol = _tol(g["x"].dtype)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], **tol)
    (out * T(g["G"])).sum().backward()
    for k, t in (("g_x", x), ("g_edge_embed", ee), ("g_a", a), ("g_a_2", a2)):
        np.testing.assert_allclose(t.grad.numpy(), g[k], atol=tol["atol"] * 20, rtol=tol["rtol"] * 50)
    if nhop is not None:
        np.testing.assert_allclose(ee2.grad.numpy(), g["g_edge_embed_nhop"], atol=tol["atol"] * 20, rtol=1e-4)


@pytest.mark.parametrize("name", GAT_CASES)
def test_gat_closed_form_backward(name):
    """The explicit gradient formulas the HIP backward kernels implement == reference autograd."""
    g = load_golden(name)
    edge = T(g["edge"])
    nhop = T(g["edge_nhop"]) if "edge_nhop" in g else None
    ee2 = T(g["edge_embed_nhop"]) if nhop is not None else None
    mask = T(g["mask"]) if float(g["train_p"]) > 0 else None
    r = O.gat_layer_backward(T(g["x"]), edge, T(g["edge_embed"]), nhop, ee2, T(g["a"]), T(g["a_2"]),
                             float(g["alpha"]), bool(g["concat"]), T(g["G"]), mask=mask)
    f64 = g["x"].dtype == np.float64
    atol = 1e-11 if f64 else 4e-4
    E1 = g["edge"].shape[1]
    np.testing.assert_allclose(r["g_x"].numpy(), g["g_x"], atol=atol, rtol=1e-4)
    np.testing.assert_allclose(r["g_edge_embed"][:E1].numpy(), g["g_edge_embed"], atol=atol, rtol=1e-4)
    np.testing.assert_allclose(r["g_a"].numpy(), g["g_a"], atol=atol, rtol=1e-4)
    np.testing.assert_allclose(r["g_a_2"].numpy(), g["g_a_2"], atol=atol, rtol=1e-4)
    if nhop is not None:
        np.testing.assert_allclose(r["g_edge_embed"][E1:].numpy(), g["g_edge_embed_nhop"], atol=atol, rtol=1e-4)


@pytest.mark.parametrize("name", ["spmm1_o1", "spmm1_oD"])
def test_spmm(name):
    g = load_golden(name)
    edge, w = T(g["edge"]), T(g["edge_w"])
    for fn in (O.spmm_rowsum, O.spmm_rowsum_aten_sequence):
        np.testing.assert_allclose(fn(edge, w, int(g["N"])).numpy(), g["out"], atol=1e-6)
    np.testing.assert_array_equal(O.spmm_rowsum_backward(edge, T(g["G"])).numpy(), g["g_edge_w"])


@pytest.mark.parametrize("name", ["spgat1_nhop", "spgat2_1hop"])
def test_spgat(name):
    g = load_golden(name)
    H = int(g["nheads"])
    leaves = {k: T(g[k]).requires_grad_(True) for k in g if k.startswith("p.")}
    x, rel = T(g["x"]).requires_grad_(True), T(g["rel"]).requires_grad_(True)
    has = "edge_nhop" in g
    y, out_rel = O.spgat_forward(
        x, rel, T(g["edge"]), T(g["edge_type"]), rel[T(g["edge_type"])],
        T(g["edge_nhop"]) if has else None, T(g["edge_type_nhop"]) if has else None,
        [leaves["p.attention_%d.a" % i] for i in range(H)], [leaves["p.attention_%d.a_2" % i] for i in range(H)],
        leaves["p.W"], leaves["p.out_att.a"], leaves["p.out_att.a_2"], float(g["alpha"]))
    np.testing.assert_allclose(y.detach().numpy(), g["out"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(out_rel.detach().numpy(), g["out_rel"], atol=2e-5, rtol=1e-5)
    ((y * T(g["G"])).sum() + (out_rel * T(g["G2"])).sum()).backward()
    np.testing.assert_allclose(x.grad.numpy(), g["g_x"], atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(rel.grad.numpy(), g["g_rel"], atol=2e-4, rtol=1e-4)
    for k, t in leaves.items():
        np.testing.assert_allclose(t.grad.numpy(), g["g." + k[2:]], atol=2e-4, rtol=1e-4, err_msg=k)


@pytest.mark.parametrize("name", ["spkbgat1_nhop", "spkbgat2_1hop"])
def test_spkbgat(name):
    g = load_golden(name)
    H = int(g["nheads"])
    P = {k[3:]: T(g[k]) for k in g if k.startswith("p0.")}
    ent = torch.nn.functional.normalize(P["entity_embeddings"], p=2, dim=1).requires_grad_(True)
    np.testing.assert_allclose(ent.detach().numpy(), g["p1.entity_embeddings"], atol=1e-7)     # the in-place normalisation
    leaves = {k: v.clone().requires_grad_(True) for k, v in P.items() if k not in ("entity_embeddings", "final_entity_embeddings", "final_relation_embeddings")}
    out_e, out_r, mask = O.spkbgat_forward(
        ent, leaves["relation_embeddings"], T(g["batch_entities"]), T(g["edge"]), T(g["edge_type"]), T(g["nhop"]),
        [leaves["sparse_gat_1.attention_%d.a" % i] for i in range(H)], [leaves["sparse_gat_1.attention_%d.a_2" % i] for i in range(H)],
        leaves["sparse_gat_1.W"], leaves["sparse_gat_1.out_att.a"], leaves["sparse_gat_1.out_att.a_2"], leaves["W_entities"], float(g["alpha"]))
    np.testing.assert_allclose(out_e.detach().numpy(), g["out_entity"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(out_r.detach().numpy(), g["out_relation"], atol=2e-5, rtol=1e-5)
    np.testing.assert_array_equal(mask.numpy(), g["mask"])
    np.testing.assert_allclose(g["p1.final_entity_embeddings"], g["out_entity"], atol=0)      # stashed by the reference forward
    ((out_e * T(g["G"])).sum() + (out_r * T(g["G2"])).sum()).backward()
    np.testing.assert_allclose(ent.grad.numpy(), g["g.entity_embeddings"], atol=2e-4, rtol=1e-4)
    for k, t in leaves.items():
        np.testing.assert_allclose(t.grad.numpy(), g["g." + k], atol=2e-4, rtol=1e-4, err_msg=k)
    # batch_test: a foreign entity table, normalised on the fly, relation table detached
    tin = torch.nn.functional.normalize(T(g["test_input"]), p=2, dim=1)
    with torch.no_grad():
        te, tr, _ = O.spkbgat_forward(
            tin, leaves["relation_embeddings"], T(g["batch_entities"]), T(g["edge"]), T(g["edge_type"]), T(g["nhop"]),
            [leaves["sparse_gat_1.attention_%d.a" % i] for i in range(H)], [leaves["sparse_gat_1.attention_%d.a_2" % i] for i in range(H)],
            leaves["sparse_gat_1.W"], leaves["sparse_gat_1.out_att.a"], leaves["sparse_gat_1.out_att.a_2"], leaves["W_entities"], float(g["alpha"]))
    np.testing.assert_allclose(te.numpy(), g["test_entity"], atol=2e-5, rtol=1e-5)
    np.testing.assert_allclose(tr.numpy(), g["test_relation"], atol=2e-5, rtol=1e-5)


def test_spkbgat_train_mode_three_sgd_iterations():
    """The regime stage A runs (GAT/main.py:478-525): drop_GAT = 0.3, train(), three iterations of forward -> batch_gat_loss -> backward ->
    SGD on three different batches; the reference's dropout factors (recorded in call order: one E-vector per head, dropout_layer on the
    concatenated heads, out_att's E-vector — GAT/models.py:71-73, :86; GAT/layers.py:158) are replayed through the oracle."""
    g = load_golden("spkbgat3_train")
    H, alpha, lr = int(g["nheads"]), float(g["alpha"]), float(g["lr"])
    P = {k[3:]: T(g[k]).clone() for k in g if k.startswith("p0.")}
    trained = [k for k in P if k not in ("final_entity_embeddings", "final_relation_embeddings")]
    for it in range(3):
        P["entity_embeddings"] = torch.nn.functional.normalize(P["entity_embeddings"], p=2, dim=1)        # :160, on .data
        leaves = {k: P[k].clone().requires_grad_(True) for k in trained}
        masks = [T(g["it%d.mask%d" % (it, h)]) for h in range(H)]
        out_e, out_r, _ = O.spkbgat_forward(
            leaves["entity_embeddings"], leaves["relation_embeddings"], T(g["it%d.batch_entities" % it]), T(g["it%d.edge" % it]),
            T(g["it%d.edge_type" % it]), T(g["it%d.nhop" % it]),
            [leaves["sparse_gat_1.attention_%d.a" % i] for i in range(H)], [leaves["sparse_gat_1.attention_%d.a_2" % i] for i in range(H)],
            leaves["sparse_gat_1.W"], leaves["sparse_gat_1.out_att.a"], leaves["sparse_gat_1.out_att.a_2"], leaves["W_entities"], alpha,
            masks=masks, layer_mask=T(g["it%d.mask%d" % (it, H)]), out_mask=T(g["it%d.mask%d" % (it, H + 1)]))
        np.testing.assert_allclose(out_e.detach().numpy(), g["it%d.out_entity" % it], atol=2e-5, rtol=1e-5)
        np.testing.assert_allclose(out_r.detach().numpy(), g["it%d.out_relation" % it], atol=2e-5, rtol=1e-5)
        loss = O.batch_gat_loss(T(g["it%d.train_indices" % it]), out_e, out_r, int(g["ratio"]), float(g["margin"]))
        np.testing.assert_allclose(loss.item(), g["losses"][it], rtol=1e-5)
        loss.backward()
        with torch.no_grad():
            for k in trained:
                P[k] = leaves[k] - lr * leaves[k].grad
    for k in trained:
        d_ref = g["p3." + k] - (g["p0." + k] if k != "entity_embeddings" else g["p3." + k] * 0 + g["p0." + k])
        np.testing.assert_allclose(P[k].numpy(), g["p3." + k], atol=1e-6, rtol=1e-5, err_msg=k)
        if k != "entity_embeddings":                                                                       # the SGD updates themselves (lr = 1e-3: tiny against the values)
            np.testing.assert_allclose(P[k].numpy() - g["p0." + k], d_ref, atol=2e-3 * np.abs(d_ref).max() + 1e-9, err_msg="delta " + k)


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


@pytest.mark.parametrize("as_gemm", [False, True])       # the reference's broadcast of matrix-vector products / one product per graph
@pytest.mark.parametrize("name", ["prop_n4d2_shared", "prop_n4d2_perbatch", "prop_n9d8_shared", "prop_n9d8_perbatch"])
def test_block_adjacency_and_propagation(name, as_gemm):
    """as_gemm is checked in float64 (it is the form the n = 32 GPU tests use as their float64 oracle): the golden vectors then
    differ from it by the REFERENCE's own fp32 round-off — gradients here are sums of terms of magnitude 1e3 with cancellation —
    hence the wider relative tolerance on that leg."""
    g = load_golden(name)
    n, d, L, B, Ts, h0, Gr, per_batch = _prop_inputs(g)
    if "T" in g:   # the small cases store T itself: the hash generator must reproduce it exactly
        np.testing.assert_array_equal(np.stack([t.numpy() for t in Ts]), g["T"])
    dt = torch.float64 if as_gemm else torch.float32
    rt = 1e-4

    def allclose(actual, desired, atol, rtol):
        if as_gemm:                                   # fp32 round-off of the reference: a few ulp of the LARGEST term of the sums
            atol, rtol = atol + 4e-6 * np.abs(desired).max(), 0.0
        np.testing.assert_allclose(actual, desired, atol=atol, rtol=rtol)
    Ts = [t.to(dt).requires_grad_(True) for t in Ts]
    ident = T(g["identity"]).to(dt).requires_grad_(True)
    h0 = h0.to(dt).requires_grad_(per_batch)
    Gr = Gr.to(dt)
    adjs = [O.build_block_adjacency(torch.relu(t), ident, n) for t in Ts]     # models/models.py:244-259
    for l in range(L):
        np.testing.assert_array_equal(adjs[l][0].detach().float().numpy(), g["adj_b0"][l])
        np.testing.assert_array_equal(adjs[l][B - 1].detach().float().numpy(), g["adj_bl"][l])
    out = O.propagate(adjs, h0, "relu", T(g["head_indices"]), T(g["tail_indices"]), as_gemm=as_gemm)
    allclose(out.detach().numpy(), g["out"], atol=1e-5, rtol=1e-5)
    (out * Gr).sum().backward()
    allclose(ident.grad.numpy(), g["g_identity"], atol=2e-3, rtol=2e-4)
    for l in range(L):
        allclose(Ts[l].grad[0].numpy(), g["g_T_b0"][l], atol=1e-4, rtol=rt)
        allclose(Ts[l].grad.sum(0).numpy(), g["g_T_sum"][l], atol=1e-3, rtol=rt)
    if per_batch:
        allclose(h0.grad[0].numpy(), g["g_h0_b0"], atol=1e-4, rtol=rt)
    np.testing.assert_array_equal(O.get_head_indices(n, d, bs=1)[0], g["head_indices"])
    np.testing.assert_array_equal(O.get_tail_indices(n, d, bs=1)[0], g["tail_indices"])
    np.testing.assert_array_equal(O.make_start_embedding(n, d).astype(np.float32), g["h0_shared"])


@pytest.mark.parametrize("act,per_batch", [("relu", True), ("tanh", False), ("linear", True)])
def test_propagate_backward_formulas_match_autograd(act, per_batch):
    """oracle.propagate_backward (the closed form the bf16 kernels are checked against, evaluated from given states) equals autograd
    of the pinned oracle.propagate in float64, arbitrary gather indices (duplicates included); with storage=bfloat16 the forward's
    states are bf16 values and a float32 re-run from them reproduces the outputs."""
    g = torch.Generator().manual_seed(5)
    B, C, S, dd, L = 3, 7, 24, 6, 3
    adjs = [((torch.rand(B, S, S, generator=g, dtype=torch.float64) - 0.45) * 0.5).requires_grad_(True) for _ in range(L)]
    h0 = (torch.randn(B, C, S, 1, generator=g, dtype=torch.float64) if per_batch else torch.randn(C, S, 1, generator=g, dtype=torch.float64)).requires_grad_(True)
    head, tail = torch.randint(0, S, (C, dd), generator=g), torch.randint(0, S, (C, dd), generator=g)
    Gr = torch.randn(B, C, L * dd, generator=g, dtype=torch.float64)
    out, states = O.propagate(adjs, h0, act, head, tail, as_gemm=True, return_states=True)
    (out * Gr).sum().backward()
    g_adj, g_h = O.propagate_backward([a.detach() for a in adjs], h0.detach(), [s_.detach() for s_ in states], act, head, tail, Gr)
    for l in range(L):
        np.testing.assert_allclose(g_adj[l].numpy(), adjs[l].grad.numpy(), atol=1e-12)
    np.testing.assert_allclose((g_h if per_batch else g_h.sum(0)).numpy(), h0.grad.squeeze(-1).numpy(), atol=1e-12)
    # storage rounding: states are exactly representable in bf16, outputs too
    out_b, st_b = O.propagate([a.detach().float() for a in adjs], h0.detach().float(), act, head, tail, as_gemm=True, storage=torch.bfloat16,
                              return_states=True)
    for s_ in st_b:
        assert torch.equal(s_.to(torch.bfloat16).float(), s_)
    assert torch.equal(out_b.to(torch.bfloat16).float(), out_b)
    np.testing.assert_allclose(out_b.numpy(), out.detach().numpy(), atol=0.05 * out.detach().abs().max().item())


def test_start_entity_embeddings():
    g = load_golden("prop3_start_entity")
    out = O.make_start_entity_embeddings(T(g["entity_embeddings"]), T(g["pos"]), int(g["d"]), T(g["template"]),
                                         max_num_nodes=int(g["n"]))
    np.testing.assert_array_equal(out.numpy(), g["out"])
    np.testing.assert_array_equal(O.get_head_indices(9, 8, bs=1)[0], g["head_indices"])
    np.testing.assert_array_equal(O.get_tail_indices(9, 8, bs=1)[0], g["tail_indices"])


@pytest.mark.parametrize("n,d", [(3, 2), (5, 3)])
def test_start_embedding_and_indices(n, d):
    g = load_golden("start_embedding_n%dd%d" % (n, d))
    np.testing.assert_array_equal(O.make_start_embedding(n, d).astype(np.float32), g["start"])
    np.testing.assert_array_equal(O.get_head_indices(n, d, bs=1)[0], g["head"])
    np.testing.assert_array_equal(O.get_tail_indices(n, d, bs=1)[0], g["tail"])


def test_linegraph_adjacency():
    g = load_golden("adj1_linegraph")
    for i, n in enumerate(range(2, 10)):
        np.testing.assert_array_equal(O.build_adjecent_matrix(n), g["adj"][i])


@pytest.mark.parametrize("name", ["gcn1_bias", "gcn1_nobias"])
def test_graph_convolution(name):
    g = load_golden(name)
    x, w = T(g["x"]).requires_grad_(True), T(g["weight"]).requires_grad_(True)
    b = T(g["bias"]).requires_grad_(True) if "bias" in g else None
    out = O.graph_convolution(x, T(g["adj"]), w, b)
    np.testing.assert_allclose(out.detach().numpy(), g["out"], atol=1e-6)
    (out * T(g["G"])).sum().backward()
    np.testing.assert_allclose(x.grad.numpy(), g["g_x"], atol=1e-6)
    np.testing.assert_allclose(w.grad.numpy(), g["g_weight"], atol=1e-5)
    if b is not None:
        np.testing.assert_allclose(b.grad.numpy(), g["g_bias"], atol=1e-5)


# ------------------------------------------------------------------------------- N1: stage-A batch builders
@pytest.mark.parametrize("name", ["sampler1_small", "sampler2_medium"])
def test_kg_batch_builders_vs_reference(name):
    """Oracle restatement of Corpus.get_graph / bfs / get_further_neighbors / get_batch_adj_data /
    get_batch_nhop_neighbors_all (GAT/create_batch.py) against the reference's own outputs: exact integer equality, in order."""
    g = load_golden(name)
    graph = O.kg_graph(T(g["adj_indices"]), T(g["adj_values"]))
    n1, n2 = O.kg_further_neighbors(graph, 1), O.kg_further_neighbors(graph, 2)
    for b in range(int(g["n_batches"])):
        ents = g["b%d_entities" % b].tolist()
        edge, etype, sset, tset = O.kg_batch_adj_data(n1, ents)
        np.testing.assert_array_equal(edge.numpy(), g["b%d_edge" % b])
        np.testing.assert_array_equal(etype.numpy(), g["b%d_edge_type" % b])
        assert sorted(sset) == g["b%d_sources" % b].tolist() and sorted(tset) == g["b%d_targets" % b].tolist()
        np.testing.assert_array_equal(O.kg_batch_nhop_neighbors(n2, ents), g["b%d_nhop" % b])
        np.testing.assert_array_equal(O.kg_batch_nhop_neighbors(n2, ents, partial_2hop=True), g["b%d_nhop_partial" % b])
    assert sum(g["b%d_nhop" % b].shape[0] for b in range(int(g["n_batches"]))) > 20        # the cases are not vacuous


@pytest.mark.parametrize("name", ["loss1_small", "loss2_wide", "loss3_ratio1"])
def test_batch_gat_loss_vs_reference(name):
    """N1: the oracle's batch_gat_loss against the reference's own function (GAT/main.py:344-376, executed by gen_golden.py): loss and
    both table gradients."""
    g = load_golden(name)
    ent = torch.from_numpy(g["entity"]).requires_grad_(True)
    rel = torch.from_numpy(g["relation"]).requires_grad_(True)
    loss = O.batch_gat_loss(torch.from_numpy(g["train_indices"]), ent, rel, int(g["ratio"]), float(g["margin"]))
    loss.backward()
    np.testing.assert_allclose(loss.detach().numpy(), g["loss"], rtol=1e-6, atol=1e-6)
    np.testing.assert_allclose(ent.grad.numpy(), g["g_entity"], rtol=1e-6, atol=1e-7)
    np.testing.assert_allclose(rel.grad.numpy(), g["g_relation"], rtol=1e-6, atol=1e-7)
