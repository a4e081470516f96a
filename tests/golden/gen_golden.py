#!/usr/bin/env python3
"""Generate the golden vectors under tests/golden/ by RUNNING THE REFERENCE (CPU).

This is the only file in the repo that imports /root/reference.  It runs in the
build container only (the reference never travels to the GPU box); its outputs
(*.npz: inputs, parameters, expected outputs, expected gradients) are committed.

Reference entry points executed (unmodified, imported from where they lie):
  GAT/layers.py:51-84    SpecialSpmmFunctionFinal / SpecialSpmmFinal
  GAT/layers.py:87-181   SpGraphAttentionLayer
  GAT/models.py:11-88    SpGAT (heads + out_att)
  models/models.py:85-277  GPGNN.forward (untied branch :238-277 = block adjacency
                           + 3-hop propagation + head*tail gather)
  models/models.py:279-487, 489-701, 703-968  RECON_EAC, RECON_EAC_KGGAT, RECON (whole models: logits + parameter gradients)
  GAT_sep_space/models.py:91-245  SpKBGATModified with W_ent2rel (+ the relation-space projection of GAT_sep_space/main.py:359-367)
  models/layers.py:35-68 GraphConvolution
  utils/build_adjecent_matrix.py:6-22
  utils/embedding_utils.py:170-202  make_start_embedding / get_head_indices / get_tail_indices
  utils/context_utils.py:387-426    make_start_entity_embeddings
  GAT/create_batch.py:391-436, 708-732, 788-895  Corpus.get_graph / bfs / get_further_neighbors /
                           get_batch_adj_data / get_batch_nhop_neighbors_all
  GAT/main.py:344-376    batch_gat_loss (the function's source is cut out of the file at generation time and executed)

Shims (live in a temp dir, never in the repo): a directory with `RECON ->
/root/reference` (models/models.py:6 imports `RECON.parsing...`), a stub `nltk`
and `tqdm` package, and the removed numpy aliases np.float / np.long.

Usage:  python tests/golden/gen_golden.py          (rewrites tests/golden/*.npz)
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

REF = "/root/reference"
OUT = os.path.dirname(os.path.abspath(__file__))


# --------------------------------------------------------------------------- shims
def _install_shims():
    shim = tempfile.mkdtemp(prefix="recon_shim_")
    os.symlink(REF, os.path.join(shim, "RECON"))
    nltk = types.ModuleType("nltk")
    nltk.word_tokenize = lambda s: s.split()
    nltk.sent_tokenize = lambda s: [s]

    class _RP:  # RegexpParser stub
        def __init__(self, *a, **k):
            pass

        def parse(self, x):
            return x
    nltk.RegexpParser = _RP
    tok = types.ModuleType("nltk.tokenize")
    tok.word_tokenize = nltk.word_tokenize
    tok.sent_tokenize = nltk.sent_tokenize
    nltk.tokenize = tok
    sys.modules["nltk"] = nltk
    sys.modules["nltk.tokenize"] = tok
    if "tqdm" not in sys.modules:
        try:
            import tqdm  # noqa: F401
        except Exception:
            tq = types.ModuleType("tqdm")
            tq.tqdm = lambda x, *a, **k: x
            sys.modules["tqdm"] = tq
    if not hasattr(np, "float"):
        np.float = float
    if not hasattr(np, "long"):
        np.long = np.int64
    sys.path.insert(0, shim)
    return shim


def hashed_uniform(shape, salt, lo=-1.0, hi=1.0):
    """Exactly reproducible pseudo-random floats (integer hash, no libm)."""
    n = int(np.prod(shape))
    i = np.arange(n, dtype=np.uint64) + np.uint64(salt) * np.uint64(0x9E3779B1)
    i = (i * np.uint64(2654435761)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(15)
    i = (i * np.uint64(2246822519)) & np.uint64(0xFFFFFFFF)
    i ^= i >> np.uint64(13)
    u = (i >> np.uint64(8)).astype(np.float64) / float(1 << 24)   # 24-bit mantissa: exact in fp32
    return (lo + (hi - lo) * u).astype(np.float32).reshape(shape)


def t2n(t):
    return t.detach().cpu().numpy()


def save(name, **arrays):
    path = os.path.join(OUT, name + ".npz")
    np.savez_compressed(path, **arrays)
    print("wrote %-28s %8.1f KB" % (name + ".npz", os.path.getsize(path) / 1024))


# --------------------------------------------------------------------------- GAT cases
def gat_case(name, gat_layers, N, F, R, D, edge, nhop=None, concat=True, dtype=torch.float32,
             train_p=0.0, alpha=0.2, seed=0):
    g = torch.Generator().manual_seed(seed + 100)
    E1 = edge.shape[1]
    x = torch.randn(N, F, generator=g, dtype=dtype)
    ee = torch.randn(E1, R, generator=g, dtype=dtype)
    if nhop is not None:
        E2 = nhop.shape[1]
        ee2 = torch.randn(E2, R, generator=g, dtype=dtype)
    else:
        E2 = 0
        nhop_t = torch.tensor([])
        ee2 = torch.tensor([])
    torch.manual_seed(seed)
    layer = gat_layers.SpGraphAttentionLayer(N, F, D, R, dropout=train_p, alpha=alpha, concat=concat)
    if dtype == torch.float64:
        layer = layer.double()
    G = torch.randn(N, D, generator=g, dtype=dtype)
    x.requires_grad_(True)
    ee.requires_grad_(True)
    if nhop is not None:
        ee2.requires_grad_(True)
    mask = np.ones(E1 + E2, dtype=np.float32)
    if train_p > 0:
        layer.train()
        torch.manual_seed(seed + 7)
        m = torch.nn.Dropout(train_p)(torch.ones(E1 + E2, dtype=dtype))
        mask = t2n(m).astype(np.float32)
        torch.manual_seed(seed + 7)     # same stream state as the mask draw above
    else:
        layer.eval()
    out = layer(x, edge, ee, nhop if nhop is not None else nhop_t, ee2)
    (out * G).sum().backward()
    arrays = dict(x=t2n(x), edge=t2n(edge), edge_embed=t2n(ee), a=t2n(layer.a), a_2=t2n(layer.a_2),
                  G=t2n(G), out=t2n(out), g_x=t2n(x.grad), g_edge_embed=t2n(ee.grad),
                  g_a=t2n(layer.a.grad), g_a_2=t2n(layer.a_2.grad), mask=mask,
                  alpha=np.float64(alpha), concat=np.int32(concat), train_p=np.float32(train_p))
    if nhop is not None:
        arrays.update(edge_nhop=t2n(nhop), edge_embed_nhop=t2n(ee2), g_edge_embed_nhop=t2n(ee2.grad))
    save(name, **arrays)


def batched_edges(B, n, e, seed=0):
    """SURVEY appendix synthetic generator (disjoint union of B graphs)."""
    g = torch.Generator().manual_seed(seed)
    base = (torch.arange(B) * n).repeat_interleave(e)
    dst = torch.randint(0, n, (B * e,), generator=g) + base
    src = torch.randint(0, n, (B * e,), generator=g) + base
    return torch.stack([dst, src])


def gen_gat():
    sys.path.insert(0, os.path.join(REF, "GAT"))
    import layers as gat_layers
    import models as gat_models
    assert gat_layers.__file__.startswith(REF)

    # GAT-1: cfg-1 shape (BASELINE.json configs[0])
    gat_case("gat1_cfg1", gat_layers, 256, 50, 50, 50, batched_edges(32, 8, 56))
    # GAT-2: duplicates, isolated nodes (9, 10, 11 never a destination), unsorted, self loops; odd dims
    g = torch.Generator().manual_seed(5)
    dst = torch.randint(0, 9, (40,), generator=g)
    src = torch.randint(0, 12, (40,), generator=g)
    dst[3], src[3] = dst[2], src[2]          # exact duplicate edge
    dst[10], src[10] = 4, 4                  # self loop
    e2 = torch.stack([dst, src])
    gat_case("gat2_dups", gat_layers, 12, 6, 5, 7, e2)
    # GAT-3: n-hop edges present
    g = torch.Generator().manual_seed(6)
    e1 = torch.randint(0, 20, (2, 30), generator=g)
    eh = torch.randint(0, 20, (2, 20), generator=g)
    gat_case("gat3_nhop", gat_layers, 20, 8, 8, 12, e1, nhop=eh)
    # GAT-4: concat=False (last layer form)
    gat_case("gat4_noconcat", gat_layers, 20, 8, 8, 12, e1, nhop=eh, concat=False)
    # GAT-5: train mode with dropout p=0.3 (mask recorded)
    gat_case("gat5_train", gat_layers, 20, 8, 8, 12, e1, nhop=eh, train_p=0.3)
    # GAT-6: fp64 twin of GAT-2
    gat_case("gat6_dups_f64", gat_layers, 12, 6, 5, 7, e2, dtype=torch.float64)
    # GAT-7: D=200 / F=R=200 slice of cfg 2 (8 graphs), vector-width-4 kernel path
    gat_case("gat7_cfg2_slice", gat_layers, 128, 200, 200, 200, batched_edges(8, 16, 64, seed=3))

    # SPMM-1
    g = torch.Generator().manual_seed(11)
    edge = torch.randint(0, 15, (2, 50), generator=g)
    for tag, od in (("o1", 1), ("oD", 9)):
        w = torch.randn(50, od, generator=g, requires_grad=True)
        G = torch.randn(15, od, generator=g)
        out = gat_layers.SpecialSpmmFinal()(edge, w, 15, 50, od)
        (out * G).sum().backward()
        save("spmm1_" + tag, edge=t2n(edge), edge_w=t2n(w), G=t2n(G), out=t2n(out), g_edge_w=t2n(w.grad),
             N=np.int32(15))

    # SpGAT-1/2: 2 heads + out_att, with / without n-hop
    for name, with_nhop in (("spgat1_nhop", True), ("spgat2_1hop", False)):
        N, nfeat, nhid, rdim, nheads, nrel = 40, 12, 8, 12, 2, 6
        g = torch.Generator().manual_seed(21)
        x = torch.randn(N, nfeat, generator=g, requires_grad=True)
        rel = torch.randn(nrel, rdim, generator=g, requires_grad=True)
        edge = torch.randint(0, N, (2, 90), generator=g)
        etype = torch.randint(0, nrel, (90,), generator=g)
        if with_nhop:
            nhop = torch.randint(0, N, (2, 35), generator=g)
            ntype = torch.randint(0, nrel, (35, 2), generator=g)
        else:
            nhop = torch.tensor([])
            ntype = torch.tensor([])
        torch.manual_seed(3)
        m = gat_models.SpGAT(N, nfeat, nhid, rdim, dropout=0.0, alpha=0.2, nheads=nheads)
        m.eval()
        ee = rel[etype]
        out, out_rel = m(None, x, rel, edge, etype, ee, nhop, ntype)
        G = torch.randn(out.shape, generator=g)
        G2 = torch.randn(out_rel.shape, generator=g)
        ((out * G).sum() + (out_rel * G2).sum()).backward()
        arrays = dict(x=t2n(x), rel=t2n(rel), edge=t2n(edge), edge_type=t2n(etype), G=t2n(G), G2=t2n(G2),
                      out=t2n(out), out_rel=t2n(out_rel), g_x=t2n(x.grad), g_rel=t2n(rel.grad),
                      nheads=np.int32(nheads), nhid=np.int32(nhid), alpha=np.float64(0.2))
        if with_nhop:
            arrays.update(edge_nhop=t2n(nhop), edge_type_nhop=t2n(ntype))
        for k, v in m.state_dict().items():
            arrays["p." + k] = t2n(v)
        for k, v in m.named_parameters():
            arrays["g." + k] = t2n(v.grad)
        save(name, **arrays)
    gen_kbgat(gat_models)
    gen_kbgat_train(gat_models)


def gen_kbgat(gat_models):
    """G7: SpKBGATModified.forward / batch_test (GAT/models.py:91-239) on a small KG: whole entity table,
    one entity batch's 1-hop + 2-hop edges, masking, W_entities skip connection, L2 normalisation."""
    N, nrel, dim, nhid, nheads = 50, 7, 10, 6, 2
    g = torch.Generator().manual_seed(31)
    ent0 = torch.randn(N, dim, generator=g)
    rel0 = torch.randn(nrel, dim, generator=g)
    edge = torch.randint(0, N, (2, 70), generator=g)
    etype = torch.randint(0, nrel, (70,), generator=g)
    nhop = torch.stack([torch.randint(0, N, (25,), generator=g), torch.randint(0, nrel, (25,), generator=g),
                        torch.randint(0, nrel, (25,), generator=g), torch.randint(0, N, (25,), generator=g)], dim=1)
    batch_entities = torch.randint(0, N, (30,), generator=g)
    for name, nh in (("spkbgat1_nhop", nhop), ("spkbgat2_1hop", torch.zeros(0, 4, dtype=torch.long))):
        torch.manual_seed(5)
        m = gat_models.SpKBGATModified(ent0.clone(), rel0.clone(), [nhid, nhid * nheads], [nhid * nheads, nhid * nheads],
                                       0.0, 0.2, [nheads, nheads], None)
        m.eval()
        sd0 = {k: t2n(v).copy() for k, v in m.state_dict().items()}
        out_e, out_r, mask = m(None, batch_entities, (edge, etype), nh)
        G = torch.randn(out_e.shape, generator=torch.Generator().manual_seed(1))
        G2 = torch.randn(out_r.shape, generator=torch.Generator().manual_seed(2))
        ((out_e * G).sum() + (out_r * G2).sum()).backward()
        with torch.no_grad():
            te, tr, tm = m.batch_test(None, batch_entities, (edge, etype), nh, ent0 * 1.7)
        arrays = dict(edge=t2n(edge), edge_type=t2n(etype), nhop=t2n(nh), batch_entities=t2n(batch_entities),
                      out_entity=t2n(out_e), out_relation=t2n(out_r), mask=t2n(mask), G=t2n(G), G2=t2n(G2),
                      test_entity=t2n(te), test_relation=t2n(tr), test_input=t2n(ent0 * 1.7),
                      nheads=np.int32(nheads), nhid=np.int32(nhid), alpha=np.float64(0.2))
        for k, v in sd0.items():
            arrays["p0." + k] = v
        for k, v in m.state_dict().items():
            arrays["p1." + k] = t2n(v)              # after forward: normalised table, stashed final_* embeddings
        for k, v in m.named_parameters():
            if v.grad is not None:
                arrays["g." + k] = t2n(v.grad)
        save(name, **arrays)


def _ref_batch_gat_loss(ratio):
    """The reference's own batch_gat_loss (GAT/main.py:344-376): main.py trains at import time, so the function's source is cut out of the
    file where it lies (ast, at generation time — nothing of it is stored) and executed against a stub `args` / CUDA = False."""
    import ast
    path = os.path.join(REF, "GAT", "main.py")
    src = open(path).read()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "batch_gat_loss")
    code = "\n".join(src.split("\n")[fn.lineno - 1:fn.end_lineno])
    ns = {"torch": torch, "args": types.SimpleNamespace(valid_invalid_ratio_gat=ratio), "CUDA": False}
    exec(compile(code, path, "exec"), ns)
    return ns["batch_gat_loss"]


def gen_kbgat_train(gat_models):
    """The regime stage A actually runs (GAT/main.py:478-525): SpKBGATModified in train() with drop_GAT = 0.3, three iterations of
    forward -> batch_gat_loss -> backward -> SGD(lr = 1e-3, the reference's default) on three DIFFERENT batches.  Every nn.Dropout of the
    model (GAT/models.py:71-73: one E-vector per head in head order, then dropout_layer on the concatenated heads, then out_att's E-vector;
    GAT/layers.py:158) draws its factors on a tensor of ones and multiplies — numerically what nn.Dropout does — so that the factors can be
    recorded in CALL ORDER and replayed by the GPU test."""
    N, nrel, dim, nhid, nheads, ratio, p_drop, lr, margin = 60, 7, 16, 8, 2, 2, 0.3, 1e-3, 1.0
    g = torch.Generator().manual_seed(41)
    ent0 = torch.randn(N, dim, generator=g)
    rel0 = torch.randn(nrel, dim, generator=g)
    torch.manual_seed(9)
    m = gat_models.SpKBGATModified(ent0.clone(), rel0.clone(), [nhid, nhid * nheads], [nhid * nheads, nhid * nheads],
                                   p_drop, 0.2, [nheads, nheads], None)
    m.train()
    arrays = {"p0." + k: t2n(v).copy() for k, v in m.state_dict().items()}
    record = []

    def recording(mod, name):
        def fwd(x):
            assert mod.training
            f = torch.nn.functional.dropout(torch.ones_like(x), mod.p, True)
            record.append((name, t2n(f).astype(np.float32)))
            return x * f
        return fwd
    names = []
    for name, mod in m.named_modules():
        if isinstance(mod, torch.nn.Dropout):
            mod.forward = recording(mod, name)
            names.append(name)
    assert sorted(names) == sorted(["sparse_gat_1.dropout_layer", "sparse_gat_1.out_att.dropout"] +
                                   ["sparse_gat_1.attention_%d.dropout" % h for h in range(nheads)]), names
    loss_fn = torch.nn.MarginRankingLoss(margin=margin)
    ref_loss = _ref_batch_gat_loss(ratio)
    opt = torch.optim.SGD(m.parameters(), lr=lr)
    losses = []
    torch.manual_seed(77)
    for it in range(3):
        E1, E2, n_pos = 80 + 7 * it, 30 + 5 * it, 20
        edge = torch.randint(0, N, (2, E1), generator=g)
        etype = torch.randint(0, nrel, (E1,), generator=g)
        nhop = torch.stack([torch.randint(0, N, (E2,), generator=g), torch.randint(0, nrel, (E2,), generator=g),
                            torch.randint(0, nrel, (E2,), generator=g), torch.randint(0, N, (E2,), generator=g)], dim=1)
        batch_entities = torch.randint(0, N, (25,), generator=g)
        pos = torch.stack([edge[1, :n_pos], etype[:n_pos], edge[0, :n_pos]], dim=1)
        neg = pos.repeat(2 * ratio, 1)
        half = neg.shape[0] // 2
        neg[:half, 0] = torch.randint(0, N, (half,), generator=g)
        neg[half:, 2] = torch.randint(0, N, (neg.shape[0] - half,), generator=g)
        tri = torch.cat([pos, neg])
        record.clear()
        out_e, out_r, mask = m(None, batch_entities, (edge, etype), nhop)
        opt.zero_grad()
        loss = ref_loss(loss_fn, tri, out_e, out_r)
        loss.backward()
        opt.step()
        losses.append(float(loss))
        assert [n for n, _ in record] == ["sparse_gat_1.attention_%d.dropout" % h for h in range(nheads)] + \
            ["sparse_gat_1.dropout_layer", "sparse_gat_1.out_att.dropout"], [n for n, _ in record]
        arrays.update({"it%d.edge" % it: t2n(edge), "it%d.edge_type" % it: t2n(etype), "it%d.nhop" % it: t2n(nhop),
                       "it%d.batch_entities" % it: t2n(batch_entities), "it%d.train_indices" % it: t2n(tri),
                       "it%d.out_entity" % it: t2n(out_e), "it%d.out_relation" % it: t2n(out_r)})
        for k, (n_, f) in enumerate(record):
            arrays["it%d.mask%d" % (it, k)] = f
    for k, v in m.state_dict().items():
        arrays["p3." + k] = t2n(v)
    save("spkbgat3_train", losses=np.asarray(losses, dtype=np.float64), nheads=np.int32(nheads), nhid=np.int32(nhid), alpha=np.float64(0.2),
         p_drop=np.float32(p_drop), lr=np.float64(lr), margin=np.float32(margin), ratio=np.int32(ratio), **arrays)


# --------------------------------------------------------------------------- GP-GNN cases
class _Fixed(torch.nn.Module):
    """Stands in for representation_to_adj[i]: returns a fixed, differentiable tensor."""

    def __init__(self, value):
        super().__init__()
        self.value = torch.nn.Parameter(value)

    def forward(self, _):
        return self.value


def gpgnn_case(name, ref_models, n, d, per_batch_h0, L=3, B=50, salt=1):
    p = {"max_num_nodes": n, "embedding_dim": d, "layer_number": L, "projection_style": "untie",
         "non-linear1": "relu", "non-linear": "tanh", "dropout1": 0.0, "position_emb": 3, "units1": 4,
         "rnn1_layers": 1, "bidirectional": 1, "batch_size": B}
    C, S, dd = n * (n - 1), 2 * d * n, (2 * d) ** 2
    emb = np.zeros((5, 3), dtype=np.float32)
    torch.manual_seed(0)
    m = ref_models.GPGNN(p, emb, max_sent_len=3, n_out=4)
    m.eval()
    Ts = [torch.from_numpy(hashed_uniform((B, C, dd), salt * 10 + i, -0.6, 1.0)) for i in range(L)]
    for i in range(L):
        m.representation_to_adj[i] = _Fixed(Ts[i].clone())
    ident = torch.from_numpy(np.eye(2 * d, dtype=np.float32) + hashed_uniform((2 * d, 2 * d), salt * 10 + 7, -0.1, 0.1))
    m.identity_transformation.data.copy_(ident)
    h0_shared = m.start_embedding.data.clone()                  # [C, S, 1]
    if per_batch_h0:                                            # RECON form, models/models.py:470
        h0 = torch.from_numpy(hashed_uniform((B, C, S, 1), salt * 10 + 8)) * h0_shared
        m.start_embedding = torch.nn.Parameter(h0.clone(), requires_grad=True)
    captured = {}
    m.linear3.register_forward_pre_hook(lambda mod, inp: captured.__setitem__("rel", inp[0]))
    real_matmul = torch.matmul
    adjs = []

    def spy_matmul(a, b):
        adjs.append(a)
        return real_matmul(a, b)
    torch.matmul = spy_matmul
    try:
        sent = torch.zeros(B, 3, dtype=torch.long)
        mark = torch.zeros(B, C, 3, dtype=torch.long)
        m(sent, mark, None)
    finally:
        torch.matmul = real_matmul
    rel = captured["rel"]                                        # [B, C, 2d*L]
    G = torch.from_numpy(hashed_uniform(tuple(rel.shape), salt * 10 + 9))
    (rel * G).sum().backward()
    arrays = dict(n=np.int32(n), d=np.int32(d), L=np.int32(L), B=np.int32(B), salt=np.int32(salt),
                  identity=t2n(ident), out=t2n(rel), G_salt=np.int32(salt * 10 + 9),
                  g_identity=t2n(m.identity_transformation.grad),
                  head_indices=t2n(m.head_indices[0]), tail_indices=t2n(m.tail_indices[0]),
                  h0_shared=t2n(h0_shared),
                  # one graph of each hop's block adjacency pins P1 without storing B*S*S floats
                  adj_b0=np.stack([t2n(a[0, 0]) for a in adjs]), adj_bl=np.stack([t2n(a[B - 1, 0]) for a in adjs]),
                  g_T_b0=np.stack([t2n(m.representation_to_adj[i].value.grad[0]) for i in range(L)]),
                  g_T_sum=np.stack([t2n(m.representation_to_adj[i].value.grad.sum(0)) for i in range(L)]))
    if per_batch_h0:
        arrays["g_h0_b0"] = t2n(m.start_embedding.grad[0])
        arrays["g_h0_sum"] = t2n(m.start_embedding.grad.sum(0))
    if B * C * dd * L * 4 < 600_000:                              # small case: store T and full grads too
        arrays["T"] = np.stack([t2n(t) for t in Ts])
        arrays["g_T"] = np.stack([t2n(m.representation_to_adj[i].value.grad) for i in range(L)])
        if per_batch_h0:
            arrays["h0"] = t2n(h0)
            arrays["g_h0"] = t2n(m.start_embedding.grad)
    save(name, **arrays)


def gpgnn_full_case(name, ref_models, style, n=3, d=2, L=3, B=50):
    """N3: the whole reference GPGNN (real embeddings, LSTM, Linear layers): state_dict, inputs, logits and the
    gradient of every trainable parameter for a fixed scalar."""
    p = {"max_num_nodes": n, "embedding_dim": d, "layer_number": L, "projection_style": style,
         "non-linear1": "relu", "non-linear": "tanh", "dropout1": 0.0, "position_emb": 3, "units1": 4,
         "rnn1_layers": 1, "bidirectional": 1, "batch_size": B}
    C = n * (n - 1)
    emb = hashed_uniform((7, 5), 301, -0.5, 0.5).astype(np.float32)
    emb[0] = 0.0
    torch.manual_seed(11)
    if style == "tie" and n != 9:
        return                                                       # the reference's tied branch hard-codes 8 x 9 (models.py:188)
    m = ref_models.GPGNN(p, emb, max_sent_len=4, n_out=3)
    m.eval()
    gs = torch.Generator().manual_seed(5)
    sent = torch.randint(1, 7, (B, 4), generator=gs)
    mark = torch.randint(0, 4, (B, C, 4), generator=gs)
    out = m(sent, mark, None)
    G = torch.from_numpy(hashed_uniform(tuple(out.shape), 302))
    (out * G).sum().backward()
    arrays = dict(n=np.int32(n), d=np.int32(d), L=np.int32(L), B=np.int32(B), style=np.array(style), emb=emb,
                  sent=t2n(sent), mark=t2n(mark), out=t2n(out), G=t2n(G))
    for k, v in m.state_dict().items():
        if k not in ("head_indices", "tail_indices", "start_embedding"):      # regenerated by the constructor; [50,C,2d] int64 is bulky
            arrays["sd." + k] = t2n(v)
    for k, v in m.named_parameters():
        if v.grad is not None:
            arrays["g." + k] = t2n(v.grad)
    save(name, **arrays)


EAC_P = {"max_num_nodes": 3, "embedding_dim": 2, "layer_number": 3, "projection_style": "untie", "non-linear1": "relu",
         "non-linear": "tanh", "dropout1": 0.0, "position_emb": 3, "units1": 4, "rnn1_layers": 1, "bidirectional": 1, "batch_size": 4,
         "char_embed_dim": 3, "hidden_dim_ent": 3, "num_entEmb_layers": 1, "is_bidirectional_ent": 1, "drop_out_rate_ent": 0.0,
         "entity_embed_dim": 2, "conv_filter_size": 2, "entity_conv_filter_size": 2, "max_char_len": 4, "char_feature_size": 3}


def recon_eac_case(name, ref_models):
    """N3: the whole reference RECON_EAC (models/models.py:279-487) — entity attribute context encoder, per-batch start
    embeddings, untied block adjacency, 3-hop propagation, classifier: state_dict, inputs, logits, parameter gradients."""
    p = dict(EAC_P)
    n, B, U, lines, wl = p["max_num_nodes"], p["batch_size"], 5, 4, 3
    C = n * (n - 1)
    emb = hashed_uniform((7, 5), 311, -0.5, 0.5).astype(np.float32)
    emb[0] = 0.0
    char_vocab = {c: i for i, c in enumerate("_abcdefg")}
    torch.manual_seed(13)
    m = ref_models.RECON_EAC(p, emb, max_sent_len=4, n_out=3, char_vocab=char_vocab)
    m.eval()
    gs = torch.Generator().manual_seed(6)
    sent = torch.randint(1, 7, (B, 4), generator=gs)
    mark = torch.randint(0, 4, (B, C, 4), generator=gs)
    ctx_words = torch.randint(0, 7, (U, lines, wl), generator=gs)
    span = p["max_char_len"] + p["conv_filter_size"] - 1
    ctx_chars = torch.randint(0, len(char_vocab), (U, lines, p["conv_filter_size"] - 1 + wl * span), generator=gs)
    mask = torch.rand(U, lines - p["entity_conv_filter_size"] + 1, generator=gs) < 0.4      # True = padding line
    mask[:, 0] = False
    pos = torch.randint(0, U, (B, C, 2), generator=gs)
    max_occ = 2
    ent = m.entity_embedding_module(ctx_words, ctx_chars, mask).detach().clone()
    out = m(sent, mark, None, None, None, ctx_words, ctx_chars, mask, pos, max_occ)
    G = torch.from_numpy(hashed_uniform(tuple(out.shape), 312))
    (out * G).sum().backward()
    arrays = dict(emb=emb, sent=t2n(sent), mark=t2n(mark), ctx_words=t2n(ctx_words), ctx_chars=t2n(ctx_chars), ctx_mask=t2n(mask),
                  pos=t2n(pos), max_occ=np.int32(max_occ), ent=t2n(ent), out=t2n(out), G=t2n(G), n_chars=np.int32(len(char_vocab)))
    for k, v in m.state_dict().items():
        if k not in ("head_indices", "tail_indices", "start_embedding"):
            arrays["sd." + k] = t2n(v)
    for k, v in m.named_parameters():
        if v.grad is not None:
            arrays["g." + k] = t2n(v.grad)
    save(name, **arrays)


KGGAT_P = dict(EAC_P, gat_entity_embedding_dim=3)


def _eac_inputs(p, m, U=5, lines=4, wl=3, seed=6):
    n, B = p["max_num_nodes"], p["batch_size"]
    C = n * (n - 1)
    gs = torch.Generator().manual_seed(seed)
    sent = torch.randint(1, 7, (B, 4), generator=gs)
    mark = torch.randint(0, 4, (B, C, 4), generator=gs)
    ctx_words = torch.randint(0, 7, (U, lines, wl), generator=gs)
    span = p["max_char_len"] + p["conv_filter_size"] - 1
    ctx_chars = torch.randint(0, 8, (U, lines, p["conv_filter_size"] - 1 + wl * span), generator=gs)
    mask = torch.rand(U, lines - p["entity_conv_filter_size"] + 1, generator=gs) < 0.4
    mask[:, 0] = False
    pos = torch.randint(0, U, (B, C, 2), generator=gs)
    return sent, mark, ctx_words, ctx_chars, mask, pos, gs


def _save_model_case(name, m, out, G, extra):
    arrays = dict(extra, out=t2n(out), G=t2n(G))
    for k, v in m.state_dict().items():
        if k not in ("head_indices", "tail_indices", "start_embedding"):
            arrays["sd." + k] = t2n(v)
    for k, v in m.named_parameters():
        if v.grad is not None:
            arrays["g." + k] = t2n(v.grad)
    save(name, **arrays)


def recon_kggat_case(name, ref_models):
    """N3 (wider): the reference RECON_EAC_KGGAT (models/models.py:489-701): RECON_EAC + the KB-GAT embeddings of every pair's
    entities concatenated in front of the classifier."""
    p = dict(KGGAT_P)
    n, B = p["max_num_nodes"], p["batch_size"]
    C = n * (n - 1)
    emb = hashed_uniform((7, 5), 321, -0.5, 0.5).astype(np.float32)
    emb[0] = 0.0
    char_vocab = {c: i for i, c in enumerate("_abcdefg")}
    torch.manual_seed(17)
    m = ref_models.RECON_EAC_KGGAT(p, emb, max_sent_len=4, n_out=3, char_vocab=char_vocab)
    m.eval()
    sent, mark, ctx_words, ctx_chars, mask, pos, gs = _eac_inputs(p, m, seed=7)
    gat = torch.randn(B, C, 2 * p["gat_entity_embedding_dim"], generator=gs)
    out = m(sent, mark, None, None, None, ctx_words, ctx_chars, mask, pos, 2, gat)
    G = torch.from_numpy(hashed_uniform(tuple(out.shape), 322))
    (out * G).sum().backward()
    _save_model_case(name, m, out, G, dict(emb=emb, sent=t2n(sent), mark=t2n(mark), ctx_words=t2n(ctx_words), ctx_chars=t2n(ctx_chars),
                                            ctx_mask=t2n(mask), pos=t2n(pos), max_occ=np.int32(2), gat=t2n(gat), n_chars=np.int32(len(char_vocab))))


def recon_full_case(name, ref_models):
    """N3 (wider): the reference's full model RECON (models/models.py:703-968): + per-relation translation residuals of the
    pairs with known KB-GAT embeddings in the relation spaces of GAT_sep_space.  The constructor writes rows of a leaf that
    requires grad (:779-785), which autograd refuses outside no_grad: built under no_grad, as a checkpoint load would."""
    p = dict(KGGAT_P)
    n, B, n_out = p["max_num_nodes"], p["batch_size"], 3
    C = n * (n - 1)
    emb = hashed_uniform((7, 5), 331, -0.5, 0.5).astype(np.float32)
    emb[0] = 0.0
    char_vocab = {c: i for i, c in enumerate("_abcdefg")}
    ent_dim, rel_dim, n_gat_rel = p["gat_entity_embedding_dim"], 4, 5
    rel_table = hashed_uniform((n_gat_rel, rel_dim), 332).astype(np.float32)
    W_all = hashed_uniform((n_gat_rel, ent_dim, rel_dim), 333).astype(np.float32)
    gat_rel = {str(i): [float(v) for v in rel_table[i]] for i in range(n_gat_rel)}
    idx2property = {0: "P0", 1: "P31", 2: "P17"}
    gat_relation2idx = {"P31": "3", "P17": "1"}                           # output 0 has no KB-GAT relation: its rows stay zero
    torch.manual_seed(19)
    saved_cuda = ref_models.CUDA
    with torch.no_grad():
        m = ref_models.RECON(p, emb, 4, n_out, char_vocab, gat_rel, W_all, idx2property, gat_relation2idx)
    m.eval()
    sent, mark, ctx_words, ctx_chars, mask, pos, gs = _eac_inputs(p, m, seed=8)
    gat = torch.randn(B, C, 2 * ent_dim, generator=gs)
    nz_pos = torch.tensor([0, 3, 4, 9, 17, 22])
    nz = torch.randn(nz_pos.numel(), 2 * ent_dim, generator=gs)
    entity_indices = torch.zeros(B, C, dtype=torch.long)                  # only its shape is read (:955-957)
    out = m(sent, mark, None, None, entity_indices, ctx_words, ctx_chars, mask, pos, 2, nz, nz_pos, gat)
    G = torch.from_numpy(hashed_uniform(tuple(out.shape), 334))
    (out * G).sum().backward()
    assert ref_models.CUDA == saved_cuda
    _save_model_case(name, m, out, G, dict(emb=emb, sent=t2n(sent), mark=t2n(mark), ctx_words=t2n(ctx_words), ctx_chars=t2n(ctx_chars),
                                            ctx_mask=t2n(mask), pos=t2n(pos), max_occ=np.int32(2), gat=t2n(gat), nz=t2n(nz), nz_pos=t2n(nz_pos),
                                            rel_table=rel_table, W_all=W_all, n_chars=np.int32(len(char_vocab))))


def gen_sep_space():
    """GAT_sep_space/models.py: SpKBGATModified with W_ent2rel — state_dict, forward, and the per-triple relation-space
    projection of GAT_sep_space/main.py:359-364 (tanh(bmm(e, W[rel]))) with its gradients."""
    sep = os.path.join(REF, "GAT_sep_space")
    for k in ("layers", "models"):
        sys.modules.pop(k, None)
    sys.path.insert(0, sep)
    try:
        import models as sep_models
        assert sep_models.__file__.startswith(sep)
        N, nrel, dim, nhid, nheads = 40, 5, 8, 4, 2
        g = torch.Generator().manual_seed(41)
        ent0, rel0 = torch.randn(N, dim, generator=g), torch.randn(nrel, dim, generator=g)
        edge, etype = torch.randint(0, N, (2, 60), generator=g), torch.randint(0, nrel, (60,), generator=g)
        nhop = torch.stack([torch.randint(0, N, (20,), generator=g), torch.randint(0, nrel, (20,), generator=g),
                            torch.randint(0, nrel, (20,), generator=g), torch.randint(0, N, (20,), generator=g)], dim=1)
        batch_entities = torch.randint(0, N, (25,), generator=g)
        triples = torch.stack([torch.randint(0, N, (33,), generator=g), torch.randint(0, nrel, (33,), generator=g),
                               torch.randint(0, N, (33,), generator=g)], dim=1)
        torch.manual_seed(7)
        m = sep_models.SpKBGATModified(ent0.clone(), rel0.clone(), [nhid, nhid * nheads], [nhid * nheads, nhid * nheads], 0.0, 0.2,
                                       [nheads, nheads], None)
        m.eval()
        sd0 = {k: t2n(v).copy() for k, v in m.state_dict().items()}
        out_e, out_r, mask = m(None, batch_entities, (edge, etype), nhop)
        W = m.W_ent2rel[triples[:, 1]]
        src = m.nonlinearity_ent2rel(torch.bmm(out_e[triples[:, 0]].unsqueeze(1), W)).squeeze()
        dst = m.nonlinearity_ent2rel(torch.bmm(out_e[triples[:, 2]].unsqueeze(1), W)).squeeze()
        x = src + out_r[triples[:, 1]] - dst                               # GAT_sep_space/main.py:366-367
        norm = torch.norm(x, p=1, dim=1)
        Gn = torch.randn(norm.shape, generator=torch.Generator().manual_seed(3))
        (norm * Gn).sum().backward()
        arrays = dict(edge=t2n(edge), edge_type=t2n(etype), nhop=t2n(nhop), batch_entities=t2n(batch_entities), triples=t2n(triples),
                      out_entity=t2n(out_e), out_relation=t2n(out_r), mask=t2n(mask), src_rel=t2n(src), dst_rel=t2n(dst), norm=t2n(norm), Gn=t2n(Gn),
                      nheads=np.int32(nheads), nhid=np.int32(nhid))
        for k, v in sd0.items():
            arrays["p0." + k] = v
        for k, v in m.named_parameters():
            if v.grad is not None:
                arrays["g." + k] = t2n(v.grad)
        save("sepspace1", **arrays)
    finally:
        sys.path.remove(sep)
        for k in ("layers", "models"):
            sys.modules.pop(k, None)


def gen_formats():
    """N4: the reference's own readers / writers (GAT/preprocess.py, GAT/main.py save_embed) on tiny synthetic files."""
    import tempfile, json as _json, importlib
    cwd = os.getcwd()
    gat = os.path.join(REF, "GAT")
    sys.path.insert(0, gat)
    try:
        pre = importlib.import_module("preprocess")
    finally:
        sys.path.remove(gat)
    assert pre.__file__.startswith(REF)
    ent_txt = "alpha 0\nbeta 1\n\ngamma 2\ndelta\t3\n"
    rel_txt = "likes 0\nknows 1\n"
    tri_txt = "alpha likes beta\nbeta knows gamma\n\ndelta likes alpha\nalpha knows gamma\n"
    e2v_txt = "0.5 -1.25 2\n1e-3 0 3.5\n"
    r2v_txt = "1 2\n-3 4.5\n"
    arrays = dict(ent_txt=np.array(ent_txt), rel_txt=np.array(rel_txt), tri_txt=np.array(tri_txt), e2v_txt=np.array(e2v_txt),
                  r2v_txt=np.array(r2v_txt))
    with tempfile.TemporaryDirectory() as d:
        paths = {}
        for name, txt in (("entity2id.txt", ent_txt), ("relation2id.txt", rel_txt), ("train.txt", tri_txt), ("entity2vec.txt", e2v_txt),
                          ("relation2vec.txt", r2v_txt)):
            paths[name] = os.path.join(d, name)
            open(paths[name], "w").write(txt)
        e2i = pre.read_entity_from_id(paths["entity2id.txt"])
        r2i = pre.read_relation_from_id(paths["relation2id.txt"])
        arrays["entity_names"] = np.array(sorted(e2i, key=e2i.get)); arrays["entity_ids"] = np.array([e2i[k] for k in sorted(e2i, key=e2i.get)])
        arrays["relation_names"] = np.array(sorted(r2i, key=r2i.get)); arrays["relation_ids"] = np.array([r2i[k] for k in sorted(r2i, key=r2i.get)])
        for tag, directed, unw in (("dir", True, False), ("undir_unw", False, True)):
            tr, (rows, cols, data), uniq = pre.load_data(paths["train.txt"], e2i, r2i, unw, directed)
            arrays["triples_" + tag] = np.array(tr); arrays["rows_" + tag] = np.array(rows); arrays["cols_" + tag] = np.array(cols)
            arrays["data_" + tag] = np.array(data); arrays["unique_" + tag] = np.array(sorted(uniq))
        ee, re_ = pre.init_embeddings(paths["entity2vec.txt"], paths["relation2vec.txt"])
        arrays["entity_emb"] = ee; arrays["relation_emb"] = re_
        # save_embed lives in GAT/main.py, which cannot be imported (argparse + datasets at import): restate its 6 lines with the
        # reference's encoder semantics (ndarray -> list) and keep the produced text as the fixture
        emb = torch.from_numpy(hashed_uniform((3, 4), 77))
        data = {idx: np.array(emb[idx]).tolist() for idx in range(emb.shape[0])}
        arrays["embed"] = t2n(emb); arrays["embed_json"] = np.array(_json.dumps(data, indent=4))
    os.chdir(cwd)
    save("formats1", **arrays)


def synthetic_kg(n_ent, n_rel, n_tri, seed):
    """Small random knowledge graph with the awkward cases on purpose: several relations on one (head, tail) pair, self
    loops, 2-cycles, entities without out-edges, targets reachable over several parents."""
    rs = np.random.RandomState(seed)
    heads = rs.randint(0, max(2, n_ent * 2 // 3), n_tri)             # the top third of the ids never appear as heads
    tails = rs.randint(0, n_ent, n_tri)
    rels = rs.randint(0, n_rel, n_tri)
    dup = rs.randint(0, n_tri, n_tri // 6)                           # repeat some (head, tail) pairs with another relation
    heads = np.concatenate([heads, heads[dup], [0, 1, 1, 2]]); tails = np.concatenate([tails, tails[dup], [0, 2, 1, 1]])
    rels = np.concatenate([rels, rs.randint(0, n_rel, dup.size), [0, 1, 2, 0]])
    return heads.astype(np.int64), rels.astype(np.int64), tails.astype(np.int64)


def gen_sampler():
    """N1: the reference's own batch builders (GAT/create_batch.py Corpus.get_graph / bfs / get_further_neighbors /
    get_batch_adj_data / get_batch_nhop_neighbors_all) on small synthetic knowledge graphs."""
    import importlib
    gat = os.path.join(REF, "GAT")
    sys.path.insert(0, gat)
    try:
        cb = importlib.import_module("create_batch")
    finally:
        sys.path.remove(gat)
    assert cb.__file__.startswith(REF)
    for name, (n_ent, n_rel, n_tri, seed) in (("sampler1_small", (12, 3, 30, 0)), ("sampler2_medium", (60, 7, 400, 1))):
        h, r, t = synthetic_kg(n_ent, n_rel, n_tri, seed)
        triples = list(zip(h.tolist(), r.tolist(), t.tolist()))
        adj = (t.tolist(), h.tolist(), r.tolist())                   # GAT/preprocess.py:73-81: rows = e2 (tail), cols = e1 (head)
        e2i = {"e%d" % i: i for i in range(n_ent)}
        r2i = {"r%d" % i: i for i in range(n_rel)}
        args = types.SimpleNamespace(entities_per_batch=5, partial_2hop=False)
        uniq = ["e%d" % i for i in range(n_ent)]
        C = cb.Corpus(args, (triples, adj), (triples[:2], adj), (triples[:2], adj), e2i, r2i, None, 8, 2, uniq, uniq, None, None, None,
                      get_2hop=True, get_1hop=True)
        arrays = dict(adj_indices=np.array([adj[0], adj[1]], dtype=np.int64), adj_values=np.array(adj[2], dtype=np.int64),
                      n_ent=np.array(n_ent))
        rs = np.random.RandomState(seed + 10)
        batches = [list(range(n_ent)), rs.permutation(n_ent)[:5].tolist(), rs.permutation(n_ent)[: max(3, n_ent // 4)].tolist(), [n_ent - 1]]
        for bi, ents in enumerate(batches):
            (ind, val), sets = C.get_batch_adj_data(None, unique_entities_train=ents, start_idx=0, end_idx=len(ents))
            arrays["b%d_entities" % bi] = np.array(ents, dtype=np.int64)
            arrays["b%d_edge" % bi] = t2n(ind).reshape(2, -1); arrays["b%d_edge_type" % bi] = t2n(val)
            arrays["b%d_sources" % bi] = np.array(sorted(sets["source"]), dtype=np.int64)
            arrays["b%d_targets" % bi] = np.array(sorted(sets["target"]), dtype=np.int64)
            arrays["b%d_nhop" % bi] = C.get_batch_nhop_neighbors_all(args, ents, C.node_neighbors_2hop).reshape(-1, 4)
            args.partial_2hop = True
            arrays["b%d_nhop_partial" % bi] = C.get_batch_nhop_neighbors_all(args, ents, C.node_neighbors_2hop).reshape(-1, 4)
            args.partial_2hop = False
        arrays["n_batches"] = np.array(len(batches))
        save(name, **arrays)


def gen_gpgnn():
    cwd = os.getcwd()
    os.chdir(REF)
    sys.path.insert(0, REF)
    try:
        from models import models as ref_models
        from models import layers as ref_layers
        from utils import build_adjecent_matrix as bam
        from utils import embedding_utils, context_utils
    finally:
        os.chdir(cwd)
    assert ref_models.__file__.startswith(REF)

    # PROP-1/2: block adjacency + 3-hop propagation, shared h0 (GPGNN form) and per-batch h0 (RECON form)
    gpgnn_full_case("gpgnn1_untied", ref_models, "untie")
    gpgnn_full_case("gpgnn2_tied_n9", ref_models, "tie", n=9, d=1, L=2)
    recon_eac_case("eac1_untied", ref_models)
    recon_kggat_case("kggat1_untied", ref_models)
    recon_full_case("recon1_untied", ref_models)
    if os.environ.get("RECON_GOLDEN_ONLY") in ("eac", "shells"):
        return
    gpgnn_case("prop_n4d2_shared", ref_models, 4, 2, per_batch_h0=False, salt=1)
    gpgnn_case("prop_n4d2_perbatch", ref_models, 4, 2, per_batch_h0=True, salt=2)
    gpgnn_case("prop_n9d8_shared", ref_models, 9, 8, per_batch_h0=False, salt=3)      # model_params.json sizes
    gpgnn_case("prop_n9d8_perbatch", ref_models, 9, 8, per_batch_h0=True, salt=4)

    # PROP-3: make_start_entity_embeddings + the index builders
    n, d, B, U = 9, 8, 3, 11
    C = n * (n - 1)
    tmpl = torch.from_numpy(embedding_utils.make_start_embedding(n, d)).float()
    ent = torch.from_numpy(hashed_uniform((U, d), 77))
    rs = np.random.RandomState(4)
    pos = torch.from_numpy(rs.randint(0, U, size=(B, C, 2)).astype(np.int64))
    vec = context_utils.make_start_entity_embeddings(ent, pos, None, d, 5, tmpl, max_num_nodes=n)
    save("prop3_start_entity", n=np.int32(n), d=np.int32(d), entity_embeddings=t2n(ent), pos=t2n(pos),
         template=t2n(tmpl), out=t2n(vec), max_occ=np.int32(5),
         head_indices=np.array(embedding_utils.get_head_indices(n, d, bs=2)[0], dtype=np.int64),
         tail_indices=np.array(embedding_utils.get_tail_indices(n, d, bs=2)[0], dtype=np.int64))
    for nn_, dd_ in ((3, 2), (5, 3)):
        save("start_embedding_n%dd%d" % (nn_, dd_),
             start=embedding_utils.make_start_embedding(nn_, dd_).astype(np.float32),
             head=np.array([list(r) for r in embedding_utils.get_head_indices(nn_, dd_, bs=1)[0]], dtype=np.int64),
             tail=np.array([list(r) for r in embedding_utils.get_tail_indices(nn_, dd_, bs=1)[0]], dtype=np.int64))

    # ADJ-1: the eight fixed 72x72 line-graph adjacencies
    save("adj1_linegraph", adj=np.stack([t2n(a) for a in bam.adjecent_matrix]))

    # GCN-1: GraphConvolution with build_adjecent_matrix(5), with and without bias
    for tag, bias in (("bias", True), ("nobias", False)):
        torch.manual_seed(9)
        layer = ref_layers.GraphConvolution(10, 6, bias=bias)
        x = torch.from_numpy(hashed_uniform((72, 10), 31)).requires_grad_(True)
        adj = bam.build_adjecent_matrix(5)
        out = layer(x, adj)
        G = torch.from_numpy(hashed_uniform((72, 6), 32))
        (out * G).sum().backward()
        arrays = dict(x=t2n(x), adj=t2n(adj), weight=t2n(layer.weight), G=t2n(G), out=t2n(out),
                      g_x=t2n(x.grad), g_weight=t2n(layer.weight.grad))
        if bias:
            arrays.update(bias=t2n(layer.bias), g_bias=t2n(layer.bias.grad))
        save("gcn1_" + tag, **arrays)


def gen_loss():
    """N1: batch_gat_loss (GAT/main.py:344-376) + nn.MarginRankingLoss.  main.py trains at import time, so the function's OWN source is
    cut out of the file where it lies (ast, at generation time — nothing of it is stored) and executed against a stub `args` / CUDA = False."""
    for name, (n_ent, n_rel, D, n_pos, ratio, margin, seed) in (("loss1_small", (9, 3, 8, 5, 2, 1.0, 0)), ("loss2_wide", (40, 6, 200, 33, 2, 5.0, 1)),
                                                               ("loss3_ratio1", (17, 4, 50, 12, 1, 0.5, 2))):
        ref_loss = _ref_batch_gat_loss(ratio)
        rs = np.random.RandomState(seed)
        ent = torch.from_numpy(hashed_uniform((n_ent, D), 90 + seed)).requires_grad_(True)
        rel = torch.from_numpy(hashed_uniform((n_rel, D), 95 + seed)).requires_grad_(True)
        pos = np.stack([rs.randint(0, n_ent, n_pos), rs.randint(0, n_rel, n_pos), rs.randint(0, n_ent, n_pos)], 1)
        neg = np.tile(pos, (2 * ratio, 1))
        half = neg.shape[0] // 2
        neg[:half, 0] = rs.randint(0, n_ent, half); neg[half:, 2] = rs.randint(0, n_ent, neg.shape[0] - half)
        neg[0] = pos[0]                                              # a pair whose two norms are equal: the term sits exactly at the margin
        tri = torch.from_numpy(np.concatenate([pos, neg]).astype(np.int64))
        loss = ref_loss(torch.nn.MarginRankingLoss(margin=margin), tri, ent, rel)
        loss.backward()
        save(name, entity=t2n(ent), relation=t2n(rel), train_indices=t2n(tri), ratio=np.int32(ratio), margin=np.float32(margin),
             loss=t2n(loss), g_entity=t2n(ent.grad), g_relation=t2n(rel.grad))


if __name__ == "__main__" and os.environ.get("RECON_GOLDEN_ONLY") == "kbgat_train":
    _install_shims()
    sys.path.insert(0, os.path.join(REF, "GAT"))
    import models as _gat_models
    assert _gat_models.__file__.startswith(REF)
    gen_kbgat_train(_gat_models)
    sys.exit(0)

if __name__ == "__main__" and os.environ.get("RECON_GOLDEN_ONLY") == "loss":
    gen_loss()
    sys.exit(0)

if __name__ == "__main__" and os.environ.get("RECON_GOLDEN_ONLY") == "formats":
    gen_formats()
    sys.exit(0)

if __name__ == "__main__" and os.environ.get("RECON_GOLDEN_ONLY") in ("eac", "shells"):
    _install_shims()
    if os.environ.get("RECON_GOLDEN_ONLY") == "shells":
        gen_sep_space()
    gen_gpgnn()
    sys.exit(0)

if __name__ == "__main__" and os.environ.get("RECON_GOLDEN_ONLY") == "sampler":
    _install_shims()
    gen_sampler()
    sys.exit(0)

if __name__ == "__main__":
    if not os.path.isdir(REF):
        sys.exit("the reference is not mounted here; golden vectors can only be regenerated in the build container")
    torch.set_num_threads(4)
    _install_shims()
    gen_gat()
    gen_sep_space()
    # GAT's `layers` / `models` module names collide with the GP-GNN package names: drop them first
    for k in ("layers", "models"):
        sys.modules.pop(k, None)
    sys.path.remove(os.path.join(REF, "GAT"))
    gen_gpgnn()
    gen_formats()
    gen_sampler()
    gen_loss()
