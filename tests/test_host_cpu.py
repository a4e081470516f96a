"""CPU-side checks (no GPU): the C-ABI library loads and exports every symbol include/recon_hip.h
declares, the host-side builders reproduce the reference's golden vectors, the drop-in modules
keep the reference's class surface, and the product path refuses CPU tensors loudly."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

from conftest import ROOT, load_golden


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "recon_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(recon_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from recon_amd import _lib
    assert os.path.exists(_lib.LIB_PATH), "build librecon_hip.so first (python -c 'import __graft_entry__ as g; g.build()')"
    h = ctypes.CDLL(_lib.LIB_PATH)
    declared = _declared_symbols()
    assert len(declared) >= 20
    for name in declared:
        assert hasattr(h, name), "librecon_hip.so does not export " + name
    bound = {s[0] for s in _lib.SYMBOLS}
    assert bound == set(declared), (bound ^ set(declared))
    L = _lib.lib()
    assert L.recon_version() == 2
    assert L.recon_error_string(-2) == b"unsupported shape"
    # size queries are pure host functions
    assert L.recon_graph_workspace_bytes(100, 1000) >= 4 * 1000 * 4
    assert L.recon_gat_bwd_partial_floats(8192, 32768, 200, 200, 200, 8) >= 256 * 1600


def test_no_cpu_fallback():
    from recon_amd.gat_layers import SpGraphAttentionLayer, SpecialSpmmFinal
    from recon_amd.gcn_layers import GraphConvolution
    from recon_amd.propagation import build_block_adjacency
    layer = SpGraphAttentionLayer(4, 3, 2, 3, 0.0, 0.2)
    with pytest.raises(RuntimeError):
        layer(torch.randn(4, 3), torch.zeros(2, 5, dtype=torch.long), torch.randn(5, 3), torch.tensor([]), torch.tensor([]))
    with pytest.raises(RuntimeError):
        SpecialSpmmFinal()(torch.zeros(2, 5, dtype=torch.long), torch.randn(5, 1), 4, 5, 1)
    with pytest.raises(RuntimeError):
        GraphConvolution(3, 2)(torch.randn(4, 3), torch.eye(4))
    with pytest.raises(RuntimeError):
        build_block_adjacency(torch.randn(1, 2, 4), torch.eye(2), 2)


def test_product_does_not_import_oracle():
    pkg = os.path.join(ROOT, "recon_amd")
    for fn in os.listdir(pkg):
        if fn.endswith(".py"):
            assert "oracle" not in open(os.path.join(pkg, fn)).read(), fn


def test_class_surface_matches_reference():
    """state_dict keys / attributes the reference's callers rely on."""
    from recon_amd import gat_layers, gcn_layers, models
    for name in ("SpGraphAttentionLayer", "SpecialSpmmFinal", "SpecialSpmmFunctionFinal", "ConvKB", "CUDA"):
        assert hasattr(gat_layers, name)
    for name in ("GraphConvolution", "SparseMM"):
        assert hasattr(gcn_layers, name)
    l = gat_layers.SpGraphAttentionLayer(10, 6, 4, 5, dropout=0.3, alpha=0.2, concat=False)
    assert list(l.state_dict().keys()) == ["a", "a_2"]
    assert l.a.shape == (4, 17) and l.a_2.shape == (1, 4)
    assert (l.in_features, l.out_features, l.num_nodes, l.alpha, l.concat, l.nrela_dim) == (6, 4, 10, 0.2, False, 5)
    assert repr(l) == "SpGraphAttentionLayer (6 -> 4)"
    g = gcn_layers.GraphConvolution(7, 3)
    assert list(g.state_dict().keys()) == ["weight", "bias"] and g.weight.shape == (7, 3)
    assert list(gcn_layers.GraphConvolution(7, 3, bias=False).state_dict().keys()) == ["weight"]
    m = models.SpGAT(10, 6, 4, 5, 0.0, 0.2, 3)
    assert sorted(m.state_dict().keys()) == sorted(["W", "out_att.a", "out_att.a_2"] +
                                                   ["attention_%d.%s" % (i, k) for i in range(3) for k in ("a", "a_2")])
    assert m.W.shape == (5, 12) and m.out_att.a.shape == (12, 36)
    ref = load_golden("spgat1_nhop")
    sd = {k[2:]: torch.from_numpy(v) for k, v in ref.items() if k.startswith("p.")}
    models.SpGAT(40, 12, 8, 12, 0.0, 0.2, 2).load_state_dict(sd, strict=True)      # the reference's own checkpoint keys
    kb = models.SpKBGATModified(torch.randn(20, 6), torch.randn(4, 6), [3, 6], [6, 6], 0.0, 0.2, [2, 2], None)
    ref_kb = load_golden("spkbgat1_nhop")
    assert sorted(kb.state_dict().keys()) == sorted(k[3:] for k in ref_kb if k.startswith("p0."))   # reference checkpoint keys
    c = gat_layers.ConvKB(8, 3, 1, 4, 0.1, 0.2)
    assert c(torch.randn(5, 24)).shape == (5, 1)
    assert set(c.state_dict().keys()) == {"conv_layer.weight", "conv_layer.bias", "fc_layer.weight", "fc_layer.bias",
                                          "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias"}


def test_sparse_mm_static_function():
    from recon_amd.gcn_layers import SparseMM
    a = torch.randn(4, 3, requires_grad=True)
    b = torch.randn(3, 5, requires_grad=True)
    out = SparseMM.apply(a, b)
    out.sum().backward()
    np.testing.assert_allclose(a.grad.numpy(), (torch.ones(4, 5) @ b.detach().t()).numpy(), rtol=1e-6)
    np.testing.assert_allclose(b.grad.numpy(), (a.detach().t() @ torch.ones(4, 5)).numpy(), rtol=1e-6)


def test_host_builders_match_reference_golden():
    from recon_amd import propagation as P
    for n, d in ((3, 2), (5, 3)):
        g = load_golden("start_embedding_n%dd%d" % (n, d))
        np.testing.assert_array_equal(P.make_start_embedding(n, d).astype(np.float32), g["start"])
        np.testing.assert_array_equal(P.get_head_indices(n, d, bs=1)[0], g["head"])
        np.testing.assert_array_equal(P.get_tail_indices(n, d, bs=1)[0], g["tail"])
    assert P.get_head_indices(9, 8).shape == (50, 72, 16)           # bs defaults to 50 as in the reference
    g = load_golden("adj1_linegraph")
    for i, n in enumerate(range(2, 10)):
        np.testing.assert_array_equal(P.build_adjecent_matrix(n).numpy(), g["adj"][i])
    g = load_golden("prop_n9d8_shared")
    np.testing.assert_array_equal(P.make_start_embedding(9, 8).astype(np.float32), g["h0_shared"])


def test_graph_cache_key_and_nhop_flag():
    from recon_amd.graph import _has_nhop
    assert not _has_nhop(torch.tensor([]))            # the reference's "no n-hop" marker: float tensor, shape [0]
    assert not _has_nhop(None)
    assert _has_nhop(torch.zeros(2, 3, dtype=torch.long))


def test_fused_head_params_alias_the_per_head_parameters():
    """SpGAT keeps attention_i.a / attention_i.a_2 (the reference's state_dict keys) as views of two fused
    buffers: the fused pair is an alias (no per-step stack), gradients reach the per-head parameters, and a
    replaced parameter storage (load / .to()) is picked up again."""
    from recon_amd.models import SpGAT
    torch.manual_seed(0)
    m = SpGAT(10, 8, 6, 4, dropout=0.0, alpha=0.2, nheads=3)
    keys = set(m.state_dict().keys())
    assert {"attention_0.a", "attention_0.a_2", "attention_2.a", "W", "out_att.a", "out_att.a_2"} <= keys
    ref_a = torch.stack([att.a.detach().clone() for att in m.attentions])
    ref_a2 = torch.cat([att.a_2.detach().clone() for att in m.attentions], dim=0)
    a, a2 = m.fused_head_params()
    assert a.shape == (3, 6, 2 * 8 + 4) and a2.shape == (3, 6)
    assert torch.equal(a, ref_a) and torch.equal(a2, ref_a2)
    assert all(att.a.data_ptr() == a[i].data_ptr() for i, att in enumerate(m.attentions))        # aliases
    (a * 2.0).sum().backward(retain_graph=True)
    (a2 * 3.0).sum().backward()
    for att in m.attentions:
        assert torch.equal(att.a.grad, torch.full_like(att.a, 2.0))
        assert torch.equal(att.a_2.grad, torch.full_like(att.a_2, 3.0))
    with torch.no_grad():
        m.attentions[1].a.add_(1.0)                                    # an optimizer's in-place update is seen
    assert torch.equal(m.fused_head_params()[0][1], ref_a[1] + 1.0)
    m.attentions[2].a.data = torch.zeros_like(m.attentions[2].a)       # storage replaced behind our back
    a_new, _ = m.fused_head_params()
    assert torch.equal(a_new[2], torch.zeros_like(ref_a[2])) and torch.equal(a_new[0], ref_a[0])
    assert m.attentions[2].a.data_ptr() == a_new[2].data_ptr()
    sd = {k: v.clone() for k, v in m.state_dict().items()}
    m2 = SpGAT(10, 8, 6, 4, dropout=0.0, alpha=0.2, nheads=3)
    m2.fused_head_params()
    m2.load_state_dict(sd)
    assert torch.equal(m2.fused_head_params()[0], a_new)


def test_synthetic_generators_match_the_oracles():
    """bench.py / tools use recon_amd.synth; the oracle keeps its own copy — same seeds, same tensors."""
    from recon_amd import synth
    from oracle import recon_oracle as O
    a = synth.synthetic_batched_graph(5, 7, 11, 6, 4, seed=3)
    b = O.synthetic_batched_graph(5, 7, 11, 6, 4, seed=3)
    for u, v in zip(a, b):
        assert torch.equal(u, v)
    g1, g2 = torch.Generator().manual_seed(5), torch.Generator().manual_seed(5)
    assert torch.equal(synth.xavier_normal((6, 10), 1.414, g1), O.xavier_normal((6, 10), 1.414, g2))


def test_only_the_allowed_places_touch_the_oracle():
    """oracle/ is test infrastructure: outside tests/ only __graft_entry__.smoke() and bench.py's baseline leg use it."""
    offenders = []
    for sub in ("recon_amd", "tools"):
        for dirpath, _, files in os.walk(os.path.join(ROOT, sub)):
            for f in files:
                if f.endswith(".py") and re.search(r"^\s*(from|import)\s+oracle\b", open(os.path.join(dirpath, f)).read(), re.M):
                    offenders.append(os.path.join(sub, f))
    assert offenders == []
    bench = open(os.path.join(ROOT, "bench.py")).read()
    assert len(re.findall(r"from oracle import", bench)) == 1 and "not args.no_cpu_baseline" in bench


@pytest.mark.parametrize("name", ["gpgnn1_untied", "gpgnn2_tied_n9"])
def test_gpgnn_state_dict_matches_reference(name):
    """the reference GPGNN's checkpoint keys and shapes, so its state_dict loads unchanged."""
    from recon_amd.gpgnn import GPGNN
    g = load_golden(name)
    p = {"max_num_nodes": int(g["n"]), "embedding_dim": int(g["d"]), "layer_number": int(g["L"]), "projection_style": str(g["style"]),
         "non-linear1": "relu", "non-linear": "tanh", "dropout1": 0.0, "position_emb": 3, "units1": 4, "rnn1_layers": 1,
         "bidirectional": 1, "batch_size": int(g["B"])}
    m = GPGNN(p, g["emb"], max_sent_len=4, n_out=3)
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    sd = m.state_dict()
    regenerated = {"head_indices", "tail_indices", "start_embedding"}
    assert set(sd.keys()) - regenerated == set(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    n, d = int(g["n"]), int(g["d"])
    assert tuple(sd["head_indices"].shape) == (50, n * (n - 1), 2 * d)          # bs = 50 baked in, models/models.py:138-142
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=False)
    assert set(missing) == regenerated and not unexpected


EAC_P = {"max_num_nodes": 3, "embedding_dim": 2, "layer_number": 3, "projection_style": "untie", "non-linear1": "relu",
         "non-linear": "tanh", "dropout1": 0.0, "position_emb": 3, "units1": 4, "rnn1_layers": 1, "bidirectional": 1, "batch_size": 4,
         "char_embed_dim": 3, "hidden_dim_ent": 3, "num_entEmb_layers": 1, "is_bidirectional_ent": 1, "drop_out_rate_ent": 0.0,
         "entity_embed_dim": 2, "conv_filter_size": 2, "entity_conv_filter_size": 2, "max_char_len": 4, "char_feature_size": 3}


KGGAT_P = dict(EAC_P, gat_entity_embedding_dim=3)


def recon_constructor_tables(g):
    """The four lookup arguments of RECON's constructor (models/models.py:708) as tests/golden/gen_golden.py::recon_full_case built them."""
    rel = {str(i): [float(v) for v in g["rel_table"][i]] for i in range(g["rel_table"].shape[0])}
    return rel, g["W_all"], {0: "P0", 1: "P31", 2: "P17"}, {"P31": "3", "P17": "1"}


def test_recon_shells_state_dict_matches_reference():
    """Wider N3: checkpoint keys and shapes of the reference's RECON_EAC_KGGAT and RECON (fixtures written by their constructors),
    and RECON's constructor-time table lookups (rows of outputs without a KB-GAT relation stay zero)."""
    from recon_amd.gpgnn import RECON_EAC_KGGAT, RECON
    regenerated = {"head_indices", "tail_indices", "start_embedding"}
    g = load_golden("kggat1_untied")
    m = RECON_EAC_KGGAT(dict(KGGAT_P), g["emb"], max_sent_len=4, n_out=3, char_vocab=list(range(int(g["n_chars"]))))
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    assert set(m.state_dict().keys()) - regenerated == set(ref.keys())
    for k, v in ref.items():
        assert tuple(m.state_dict()[k].shape) == tuple(v.shape), k
    assert m.linear3.in_features == 2 * 2 * 3 + 2 * 3
    g = load_golden("recon1_untied")
    m = RECON(dict(KGGAT_P), g["emb"], 4, 3, list(range(int(g["n_chars"]))), *recon_constructor_tables(g))
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    assert set(m.state_dict().keys()) - {"start_embedding"} == set(ref.keys())           # no head_indices / tail_indices keys in this class
    for k, v in ref.items():
        assert tuple(m.state_dict()[k].shape) == tuple(v.shape), k
    np.testing.assert_array_equal(m.gat_relation_embeddings.detach().numpy(), ref["gat_relation_embeddings"])
    np.testing.assert_array_equal(m.W_ent2rel.numpy(), ref["W_ent2rel"])
    assert not m.gat_relation_embeddings[0].any() and m.gat_relation_embeddings.requires_grad and not m.W_ent2rel.requires_grad
    assert m.linear3.in_features == 2 * 2 * 3 + 2 * 3 + 3


def test_sep_space_state_dict_matches_reference():
    """GAT_sep_space/models.py:91-245: the GAT tree's keys + W_ent2rel; the reference's own checkpoint loads with strict=True."""
    from recon_amd.sep_space import SpKBGATModified
    g = load_golden("sepspace1")
    sd0 = {k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("p0.")}
    H, nhid = int(g["nheads"]), int(g["nhid"])
    m = SpKBGATModified(sd0["entity_embeddings"].clone(), sd0["relation_embeddings"].clone(), [nhid, nhid * H], [nhid * H, nhid * H], 0.0, 0.2, [H, H], None)
    assert sorted(m.state_dict().keys()) == sorted(sd0.keys())
    m.load_state_dict(sd0, strict=True)
    assert m.W_ent2rel.shape == (sd0["relation_embeddings"].shape[0], nhid * H, nhid * H) and m.nonlinearity_ent2rel is torch.tanh


def test_recon_eac_state_dict_matches_reference():
    """the reference RECON_EAC's checkpoint keys and shapes (fixture written by running its constructor), and
    its entity-context encoder — stock ops, so it runs here — against the reference's entity vectors (the numerical check of
    the whole model is the GPU test)."""
    from recon_amd.gpgnn import RECON_EAC
    g = load_golden("eac1_untied")
    m = RECON_EAC(dict(EAC_P), g["emb"], max_sent_len=4, n_out=3, char_vocab=list(range(int(g["n_chars"]))))
    ref = {k[3:]: v for k, v in g.items() if k.startswith("sd.")}
    sd = m.state_dict()
    regenerated = {"head_indices", "tail_indices", "start_embedding"}
    assert set(sd.keys()) - regenerated == set(ref.keys())
    for k, v in ref.items():
        assert tuple(sd[k].shape) == tuple(v.shape), k
    assert tuple(sd["head_indices"].shape) == (EAC_P["batch_size"], 6, 4)          # bs = p['batch_size'], models/models.py:338-342
    missing, unexpected = m.load_state_dict({k: torch.from_numpy(np.asarray(v)) for k, v in ref.items()}, strict=False)
    assert set(missing) == regenerated and not unexpected
    assert m.entity_embedding_module.word_embeddings is m.word_embedding           # one table, two keys
    ent = m.eval().entity_embedding_module(torch.from_numpy(g["ctx_words"]), torch.from_numpy(g["ctx_chars"]), torch.from_numpy(g["ctx_mask"]))
    np.testing.assert_allclose(ent.detach().numpy(), g["ent"], atol=1e-6)


def test_on_disk_formats_match_reference_readers(tmp_path):
    """id maps, triple files, embedding text files parsed exactly like GAT/preprocess.py does (fixture:
    the reference's own functions on tiny synthetic files), and the final_*_embeddings.json / W_ent2rel round trips."""
    from recon_amd import formats
    g = load_golden("formats1")
    files = {"entity2id.txt": "ent_txt", "relation2id.txt": "rel_txt", "train.txt": "tri_txt", "entity2vec.txt": "e2v_txt",
             "relation2vec.txt": "r2v_txt"}
    for fn, key in files.items():
        (tmp_path / fn).write_text(str(g[key]))
    e2i = formats.read_entity_from_id(str(tmp_path / "entity2id.txt"))
    r2i = formats.read_relation_from_id(str(tmp_path / "relation2id.txt"))
    assert e2i == dict(zip([str(s) for s in g["entity_names"]], [int(v) for v in g["entity_ids"]]))
    assert r2i == dict(zip([str(s) for s in g["relation_names"]], [int(v) for v in g["relation_ids"]]))
    for tag, directed, unw in (("dir", True, False), ("undir_unw", False, True)):
        tr, (rows, cols, data), uniq = formats.load_data(str(tmp_path / "train.txt"), e2i, r2i, unw, directed)
        np.testing.assert_array_equal(np.array(tr), g["triples_" + tag])
        np.testing.assert_array_equal(np.array(rows), g["rows_" + tag])
        np.testing.assert_array_equal(np.array(cols), g["cols_" + tag])
        np.testing.assert_array_equal(np.array(data), g["data_" + tag])
        assert sorted(uniq) == [str(s) for s in g["unique_" + tag]]
        edge, etype = formats.edges_from_adjacency((rows, cols, data))
        assert edge.shape == (2, len(rows)) and edge.dtype == torch.int64 and etype.tolist() == list(data)
    ee, re_ = formats.init_embeddings(str(tmp_path / "entity2vec.txt"), str(tmp_path / "relation2vec.txt"))
    np.testing.assert_array_equal(ee, g["entity_emb"])
    np.testing.assert_array_equal(re_, g["relation_emb"])
    formats.save_embed(torch.from_numpy(g["embed"]), str(tmp_path / "final_entity_embeddings.json"))
    assert (tmp_path / "final_entity_embeddings.json").read_text() == str(g["embed_json"])
    np.testing.assert_array_equal(formats.load_embed(str(tmp_path / "final_entity_embeddings.json")), g["embed"])
    formats.save_w_ent2rel(torch.from_numpy(g["embed"]), str(tmp_path))
    assert (tmp_path / "W_ent2rel.json.npy").exists()
    np.testing.assert_array_equal(formats.load_w_ent2rel(str(tmp_path)), g["embed"])


def test_gemm_family_policy(monkeypatch):
    """auto: small products stay on the exact-fp32 MFMA GEMMs (None workspace); 0: always."""
    from recon_amd import gat_layers
    monkeypatch.setattr(gat_layers, "_GEMM_BX3", "auto")
    assert gat_layers._atp_split_buffer(50, 50, 50, 1, "cpu", N=256) == (None, None)   # cfg 1
    assert 2.0 * 8192 * 600 * 8 * 200 > gat_layers._BX3_MIN_FLOP                          # cfg 2 takes the split-precision kernels
    monkeypatch.setattr(gat_layers, "_GEMM_BX3", "0")
    assert gat_layers._atp_split_buffer(200, 200, 200, 8, "cpu", N=8192) == (None, None)


def test_propagation_workspace_queries():
    """Host-side size queries of the propagation entry points (no device work): the wide-state forms exist for 160 < S <= 512 only, their
    workspaces are sized per slice of at most 256 graphs / per batch, and block mode is accepted where its shape identities hold."""
    import ctypes as C
    from recon_amd import _lib
    L = _lib.lib()

    def args(B, Cn, S, Lh, dd, trans=False):
        a = _lib.PropArgs(B, Cn, S, Lh, dd, 1, None, None, 0, None, None, 0, None, None, None, None, None, None, 0)
        if trans:
            a.trans = (C.c_void_p * Lh)()
        return a
    assert L.recon_propagate_ws_bytes(C.byref(args(1024, 72, 144, 3, 16))) == 0            # whole graph per workgroup: no workspace
    assert L.recon_propagate_ws_bytes(C.byref(args(1024, 992, 516, 3, 16))) == 0           # wider than the form takes
    one = L.recon_propagate_ws_bytes(C.byref(args(1, 992, 512, 3, 16)))
    full = L.recon_propagate_ws_bytes(C.byref(args(1024, 992, 512, 3, 16)))
    assert one >= 3 * 512 * 512 * 4 and full < 257 * one and full > 200 * one              # 256-graph slices
    assert L.recon_propagate_ws_bytes(C.byref(args(4, 992, 512, 3, 16, trans=True))) > 0   # n = 32 blocks: S = 16 n, C = n (n - 1)
    assert L.recon_propagate_ws_bytes(C.byref(args(4, 990, 512, 3, 16, trans=True))) == 0
    assert L.recon_propagate_bwd_ws_floats(C.byref(args(7, 992, 512, 3, 16))) == 7 * 992 * 512
    assert L.recon_propagate_bwd_ws_floats(C.byref(args(7, 72, 144, 3, 16))) == 0


def test_trust_marks_carry_version_and_bound():
    """graph.trust(): a mark is honoured only while the tensor is unmodified and only by a consumer whose table is at least as large as
    the bound the values were validated against (advisor, round 3: a permanent object tag let an edited tensor, or a smaller entity table,
    skip the only range check)."""
    import torch
    from recon_amd.graph import trust, trusted, trust_bounds
    t = torch.arange(10)
    assert not trusted(t)
    trust(t, bound=10)
    assert trusted(t) and trusted(t, 10) and trusted(t, 50)
    assert not trusted(t, 9)                                              # a table of 9 rows: values up to 9 were allowed
    t[3] = 1000                                                           # an in-place edit bumps the version: the mark is void
    assert not trusted(t)
    q = trust(torch.zeros(4, 4, dtype=torch.int64), bound=7, rel_bound=3)  # mixed ids (the 2-hop quadruples): both bounds handed down
    assert trust_bounds(q) == (7, 3) and trust_bounds(torch.zeros(1)) == (None, None)
    k = trust(torch.arange(5))                                            # keys built for exactly the table they index: no bound
    assert trusted(k, 5) and trusted(k, 1)
