"""Randomised parity sweeps (fixed seeds): shapes, degree distributions, GEMM families, table / materialised edge embeddings and hub
splitting drawn at random, each case against the oracle or against the plainest configuration of the same kernels.  These sweeps
found, in round 2: the 64 KiB LDS stages missing from recon_gat_atp_supported() (wide inputs x many heads failed instead of falling
back) and SpecialSpmmFinal's view of an empty edge list."""
import numpy as np
import pytest
import torch

from oracle import recon_oracle as O
from test_gat_gpu import dev

pytestmark = pytest.mark.gpu


class _Checker:
    def __init__(self):
        self.bad = []

    def __call__(self, tag, cfg, actual, desired, atol=1e-4, rel=1e-4):
        a = actual.detach().cpu().double()
        b = desired.detach().cpu().double()
        tol = atol + rel * (float(b.abs().max()) if b.numel() else 0.0)
        err = float((a - b).abs().max()) if b.numel() else 0.0
        if not err <= tol or not bool(torch.isfinite(a).all()):
            self.bad.append((tag, cfg, err, tol))


def test_fuzz_gat_heads_hubs_and_table_vs_plain_path(monkeypatch):
    """150 random layer calls: hub rows cut into pieces + indexed relation table + a random GEMM family, against the same call with
    one wave per row, materialised edge embeddings and the exact-fp32 GEMMs (which the other tests pin to the oracle)."""
    from recon_amd import gat_layers, graph as graph_mod
    d = dev()
    rs = np.random.RandomState(0)
    chk = _Checker()

    def run(edge, N, x, table, index, a, a2, keep, G, concat, chunk, use_table, fam):
        monkeypatch.setattr(graph_mod, "HUB_CHUNK", chunk)
        monkeypatch.setattr(gat_layers, "_GEMM_BX3", fam)
        graph_mod.clear_graph_cache()
        gr = graph_mod.prepare_graph(edge.to(d), None, N)
        xd, td, ad, a2d = (t.to(d).requires_grad_(True) for t in (x, table, a, a2))
        kd = keep.to(d) if keep is not None else None
        if use_table:
            out = gat_layers.gat_heads(xd, td, ad, a2d, gr, kd, 0.2, concat, ee_index=index.to(d))
        else:
            out = gat_layers.gat_heads(xd, gat_layers.gather_rows(td, index.to(d)), ad, a2d, gr, kd, 0.2, concat)
        (out * G.to(d)).sum().backward()
        return [t.detach().cpu() for t in (out, xd.grad, ad.grad, a2d.grad, td.grad)]

    for it in range(150):
        N = int(rs.randint(1, 400)); E = int(rs.choice([0, 1, 5, 70, 300, 2000, 6000]))
        F_ = int(rs.choice([2, 4, 6, 8, 10, 16, 24, 50, 64, 100, 200, 264, 520, 1040]))
        R = int(rs.choice([2, 4, 8, 10, 16, 50, 64, 200, 300, 1600]))
        D = int(rs.choice([8, 16, 24, 25, 40, 64])); H = int(rs.randint(1, 10))
        concat = bool(rs.randint(0, 2)); drop = bool(rs.randint(0, 2))
        nrel = int(rs.randint(1, 12)); extra = int(rs.choice([0, 0, min(E, 30)]))
        fam = str(rs.choice(["2", "1", "0", "auto"]))
        cfg = (it, N, E, F_, R, D, H, concat, drop, nrel, extra, fam)
        g = torch.Generator().manual_seed(it)
        if E:
            p = 1.0 / np.arange(1, N + 1) ** rs.choice([0.0, 1.0, 1.5]); p /= p.sum()
            dst = torch.from_numpy(rs.choice(N, size=E, p=p)); src = torch.from_numpy(rs.choice(N, size=E, p=p[::-1].copy()))
        else:
            dst = torch.zeros(0, dtype=torch.long); src = torch.zeros(0, dtype=torch.long)
        edge = torch.stack([dst, src]).long()
        table = torch.randn(nrel + extra, R, generator=g) * 0.5
        index = torch.randint(0, nrel, (E,), generator=g)
        if extra:
            index[E - extra:] = nrel + torch.arange(extra)
        x = torch.randn(N, F_, generator=g)
        a = torch.randn(H, D, 2 * F_ + R, generator=g) * (1.0 / np.sqrt(2 * F_ + R)); a2 = torch.randn(H, D, generator=g) * 0.3
        keep = (torch.rand(H, E, generator=g) > 0.3).float() / 0.7 if drop and E else None
        G = torch.randn(N, H * D, generator=g)
        ref = run(edge, N, x, table, index, a, a2, keep, G, concat, 0, False, "0")
        new = run(edge, N, x, table, index, a, a2, keep, G, concat, 64, True, fam)
        for nm, r_, n_ in zip(("out", "g_x", "g_a", "g_a_2", "g_table"), ref, new):
            chk(nm, cfg, n_, r_)
    graph_mod.clear_graph_cache()
    assert not chk.bad, chk.bad[:5]


def test_fuzz_spgat_model_vs_oracle(monkeypatch):
    """40 random SpGAT models (1-hop + 2-hop edges, skewed degrees, 1-8 heads, edge_embed passed or None, random GEMM family):
    both outputs and every gradient against the oracle's float64 restatement of GAT/models.py:47-88."""
    from recon_amd.models import SpGAT
    from recon_amd import gat_layers, graph as graph_mod
    d = dev()
    rs = np.random.RandomState(7)
    chk = _Checker()
    for it in range(40):
        g = torch.Generator().manual_seed(2000 + it)
        N = int(rs.choice([5, 40, 300, 1500])); E1 = int(rs.choice([0, 10, 400, 6000])); E2 = int(rs.choice([0, 0, 7, 900]))
        F_ = int(rs.choice([4, 10, 50, 100])); D = int(rs.choice([8, 25, 50, 100])); H = int(rs.choice([1, 2, 4, 8])); nrel = int(rs.randint(1, 30))
        fam = str(rs.choice(["2", "1", "0", "auto"])); dense = bool(rs.randint(0, 2))
        cfg = (it, N, E1, E2, F_, D, H, nrel, fam, dense)
        monkeypatch.setattr(gat_layers, "_GEMM_BX3", fam)
        graph_mod.clear_graph_cache()
        p = 1.0 / np.arange(1, N + 1) ** rs.choice([0.0, 1.0]); p /= p.sum()

        def edges(E):
            if E == 0:
                return torch.zeros(2, 0, dtype=torch.long)
            return torch.from_numpy(np.stack([rs.choice(N, size=E, p=p), rs.choice(N, size=E, p=p[::-1].copy())])).long()
        edge = edges(E1); et = torch.randint(0, nrel, (E1,), generator=g)
        edge_nhop = edges(E2) if E2 else torch.tensor([])
        et_nhop = torch.randint(0, nrel, (E2, 2), generator=g) if E2 else torch.tensor([])
        x = torch.randn(N, F_, generator=g); rel = torch.randn(nrel, F_, generator=g)
        torch.manual_seed(it)
        m = SpGAT(N, F_, D, F_, 0.0, 0.2, H)
        pr = dict(head_a=[a.a.detach().clone().double().requires_grad_(True) for a in m.attentions],
                  head_a2=[a.a_2.detach().clone().double().requires_grad_(True) for a in m.attentions],
                  W=m.W.detach().clone().double().requires_grad_(True), out_a=m.out_att.a.detach().clone().double().requires_grad_(True),
                  out_a2=m.out_att.a_2.detach().clone().double().requires_grad_(True))
        m = m.to(d)
        xd, reld, etd = x.to(d).requires_grad_(True), rel.to(d).requires_grad_(True), et.to(d)
        ee = gat_layers.gather_rows(reld, etd) if dense else None
        y, orel = m(None, xd, reld, edge.to(d), etd, ee, edge_nhop.to(d) if E2 else edge_nhop, et_nhop.to(d) if E2 else et_nhop)
        Gy = torch.randn(N, H * D, generator=g); Gr = torch.randn(nrel, H * D, generator=g)
        ((y * Gy.to(d)).sum() + (orel * Gr.to(d)).sum()).backward()
        xr, relr = x.double().requires_grad_(True), rel.double().requires_grad_(True)
        yr, orr = O.spgat_forward(xr, relr, edge, et, relr[et], edge_nhop if E2 else None, et_nhop if E2 else None, pr["head_a"], pr["head_a2"],
                                  pr["W"], pr["out_a"], pr["out_a2"], 0.2)
        ((yr * Gy.double()).sum() + (orr * Gr.double()).sum()).backward()
        chk("y", cfg, y, yr); chk("out_rel", cfg, orel, orr); chk("g_x", cfg, xd.grad, xr.grad); chk("g_rel", cfg, reld.grad, relr.grad)
        chk("g_W", cfg, m.W.grad, pr["W"].grad); chk("g_out_a", cfg, m.out_att.a.grad, pr["out_a"].grad)
        chk("g_out_a_2", cfg, m.out_att.a_2.grad, pr["out_a2"].grad)
        for h in range(H):
            chk("g_a[%d]" % h, cfg, m.attentions[h].a.grad, pr["head_a"][h].grad)
            chk("g_a_2[%d]" % h, cfg, m.attentions[h].a_2.grad, pr["head_a2"][h].grad)
    graph_mod.clear_graph_cache()
    assert not chk.bad, chk.bad[:5]


def test_fuzz_propagation_gcn_rowsum_small_mm_vs_oracle():
    """40 random cases each of: block adjacency + propagation (n, d, hops, batch, non-linearity, shared / per-batch start state),
    GraphConvolution (2-D and batched, n across the 32-row tile edges), SpecialSpmmFinal (empty to 40 k edges, skewed), small_mm."""
    from recon_amd.propagation import build_block_adjacency, propagate, make_start_embedding, get_head_indices, get_tail_indices
    from recon_amd.gcn_layers import GraphConvolution
    from recon_amd.gat_layers import SpecialSpmmFinal, small_mm
    d_ = dev()
    rs = np.random.RandomState(3)
    chk = _Checker()
    for it in range(40):
        g = torch.Generator().manual_seed(1000 + it)
        n = int(rs.randint(2, 10)); d = int(rs.choice([1, 2, 3, 4, 8])); L = int(rs.randint(1, 5)); B = int(rs.choice([1, 2, 5, 17, 50]))
        act = str(rs.choice(["relu", "tanh", "linear"])); per_batch = bool(rs.randint(0, 2))
        cfg = ("prop", it, n, d, L, B, act, per_batch)
        C, S, dd = n * (n - 1), 2 * d * n, 2 * d
        Ts = [(torch.rand(B, C, dd * dd, generator=g) - 0.4) * 0.5 for _ in range(L)]
        ident = torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)
        tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
        h0 = (torch.randn(B, C, S, 1, generator=g) * tmpl) if per_batch else tmpl
        head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]); tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
        Gr = torch.randn(B, C, dd * L, generator=g)

        def run(device, build, prop):
            Tl = [t.clone().to(device).requires_grad_(True) for t in Ts]
            I = ident.clone().to(device).requires_grad_(True)
            h = h0.clone().to(device).requires_grad_(per_batch)
            out = prop([build(torch.relu(t), I, n) for t in Tl], h, act, head.to(device), tail.to(device))
            (out * Gr.to(device)).sum().backward()
            return [out] + [t.grad for t in Tl] + [I.grad] + ([h.grad] if per_batch else [])
        for k, (x_, y_) in enumerate(zip(run(d_, build_block_adjacency, propagate), run("cpu", O.build_block_adjacency, O.propagate))):
            chk("prop[%d]" % k, cfg, x_, y_)

        B = int(rs.choice([1, 3, 20])); n = int(rs.choice([1, 2, 9, 31, 32, 33, 70, 130])); I_ = int(rs.choice([1, 3, 8, 50, 300]))
        O_ = int(rs.choice([1, 5, 64, 300])); two_d = bool(rs.randint(0, 4) == 0); bias = bool(rs.randint(0, 2))
        cfg = ("gcn", it, B, n, I_, O_, two_d, bias)
        x = torch.randn(B, n, I_, generator=g); adj = (torch.rand(B, n, n, generator=g) < 0.3).float() + torch.eye(n)
        adj = adj / adj.sum(-1, keepdim=True)
        if two_d:
            x, adj = x[0], adj[0]
        torch.manual_seed(it)
        layer = GraphConvolution(I_, O_, bias=bias)
        w = layer.weight.detach().clone(); b = layer.bias.detach().clone() if bias else None
        Gr = torch.randn(*x.shape[:-1], O_, generator=g)
        xr, adjr, wr = (t.clone().requires_grad_(True) for t in (x, adj, w)); br = b.clone().requires_grad_(True) if bias else None
        ref = O.graph_convolution(xr, adjr, wr, br); (ref * Gr).sum().backward()
        layer = layer.to(d_); xd, adjd = x.to(d_).requires_grad_(True), adj.to(d_).requires_grad_(True)
        out = layer(xd, adjd); (out * Gr.to(d_)).sum().backward()
        chk("gcn out", cfg, out, ref); chk("gcn g_x", cfg, xd.grad, xr.grad); chk("gcn g_adj", cfg, adjd.grad, adjr.grad)
        chk("gcn g_weight", cfg, layer.weight.grad, wr.grad)
        if bias:
            chk("gcn g_bias", cfg, layer.bias.grad, br.grad)

        N = int(rs.randint(1, 500)); E = int(rs.choice([0, 1, 7, 300, 5000, 40000])); C_ = int(rs.choice([1, 2, 3, 4, 25, 50, 200]))
        cfg = ("spmm", it, N, E, C_)
        p = 1.0 / np.arange(1, N + 1) ** rs.choice([0.0, 1.0, 2.0]); p /= p.sum()
        dst = torch.from_numpy(rs.choice(N, size=E, p=p)).long() if E else torch.zeros(0, dtype=torch.long)
        edge = torch.stack([dst, torch.zeros_like(dst)]); w = torch.randn(E, C_, generator=g)
        wd = w.to(d_).requires_grad_(True)
        out = SpecialSpmmFinal()(edge.to(d_), wd, N, E, C_)
        chk("spmm", cfg, out, O.spmm_rowsum(edge, w.double(), N), atol=1e-5, rel=3e-6)
        Gr = torch.randn(N, C_, generator=g); (out * Gr.to(d_)).sum().backward()
        chk("spmm backward", cfg, wd.grad, Gr[dst] if E else torch.zeros(0, C_), atol=0, rel=0)

        M = int(rs.choice([1, 3, 64, 237, 3000, 20000])); K = int(rs.choice([1, 5, 50, 100, 200, 1030, 1600])); N2 = int(rs.choice([1, 7, 50, 200, 1600]))
        cfg = ("mm", it, M, K, N2)
        A = torch.randn(M, K, generator=g); Bm = torch.randn(K, N2, generator=g); Gr = torch.randn(M, N2, generator=g)
        Ad, Bd = A.to(d_).requires_grad_(True), Bm.to(d_).requires_grad_(True)
        out = small_mm(Ad, Bd); (out * Gr.to(d_)).sum().backward()
        chk("mm", cfg, out, A.double() @ Bm.double(), rel=2e-5); chk("mm g_A", cfg, Ad.grad, Gr.double() @ Bm.double().t(), rel=2e-5)
        chk("mm g_B", cfg, Bd.grad, A.double().t() @ Gr.double(), rel=2e-5)
    assert not chk.bad, chk.bad[:5]


def test_fuzz_formulations_inference_and_spkbgat_vs_oracle(monkeypatch):
    """40 random cases each of: the two formulations of the layer against each other (aggregate-then-project vs project-then-aggregate:
    outputs and all gradients) and the inference call against the training call; SpKBGATModified.forward and .batch_test (whole entity
    table, 1-hop + 2-hop edges, mask, W_entities skip, L2 norm) against the oracle's restatement of GAT/models.py:136-239."""
    from recon_amd import gat_layers, graph as graph_mod
    from recon_amd.models import SpKBGATModified
    d = dev()
    rs = np.random.RandomState(0)
    chk = _Checker()
    for it in range(40):
        g = torch.Generator().manual_seed(it)
        # ---- proj path vs atp path, and eval (no grad) vs train outputs
        N = int(rs.randint(1, 300)); E = int(rs.choice([0, 3, 200, 3000])); F_ = int(rs.choice([3, 4, 7, 16, 50])); R = int(rs.choice([1, 4, 5, 16, 50]))
        D = int(rs.choice([1, 8, 25, 40])); H = int(rs.randint(1, 6)); concat = bool(rs.randint(0, 2))
        cfg = ("paths", it, N, E, F_, R, D, H, concat)
        try:
            p = 1.0 / np.arange(1, N + 1); p /= p.sum()
            edge = torch.from_numpy(np.stack([rs.choice(N, size=E, p=p), rs.randint(0, N, size=E)])).long() if E else torch.zeros(2, 0, dtype=torch.long)
            x = torch.randn(N, F_, generator=g); ee = torch.randn(E, R, generator=g) * 0.5
            a = torch.randn(H, D, 2 * F_ + R, generator=g) / np.sqrt(2 * F_ + R); a2 = torch.randn(H, D, generator=g) * 0.3
            G = torch.randn(N, H * D, generator=g)
            res = {}
            for path in ("atp", "proj"):
                monkeypatch.setattr(gat_layers, "_GAT_PATH", path); monkeypatch.setattr(gat_layers, "_GEMM_BX3", "auto")
                graph_mod.clear_graph_cache()
                gr = graph_mod.prepare_graph(edge.to(d), None, N)
                xd, eed, ad, a2d = (t.to(d).requires_grad_(True) for t in (x, ee, a, a2))
                out = gat_layers.gat_heads(xd, eed, ad, a2d, gr, None, 0.2, concat)
                (out * G.to(d)).sum().backward()
                with torch.no_grad():
                    oi = gat_layers.gat_heads(x.to(d), ee.to(d), a.to(d), a2.to(d), gr, None, 0.2, concat)
                chk("eval " + path, cfg, oi, out, atol=1e-6, rel=1e-6)
                res[path] = [out, xd.grad, eed.grad, ad.grad, a2d.grad]
            for nm, u, v in zip(("out", "g_x", "g_ee", "g_a", "g_a2"), res["atp"], res["proj"]):
                chk("atp vs proj " + nm, cfg, u, v)
            monkeypatch.setattr(gat_layers, "_GAT_PATH", "auto")
        except Exception as ex:
            chk.bad.append(("exception", cfg, repr(ex)[:300], 0))
        # ---- SpKBGATModified forward / batch_test vs oracle
        Ne = int(rs.choice([20, 200, 1000])); E1 = int(rs.choice([0, 50, 3000])); E2 = int(rs.choice([0, 30, 500])); nrel = int(rs.randint(1, 20)); emb = int(rs.choice([4, 10, 50]))
        d1 = int(rs.choice([8, 25, 100])); h1 = int(rs.choice([1, 2, 4]))
        cfg = ("kbgat", it, Ne, E1, E2, nrel, emb, d1, h1)
        try:
            graph_mod.clear_graph_cache()
            edge = torch.from_numpy(np.stack([rs.randint(0, Ne, E1), rs.randint(0, Ne, E1)])).long()
            et = torch.randint(0, nrel, (E1,), generator=g)
            quads = torch.stack([torch.randint(0, Ne, (E2,), generator=g), torch.randint(0, nrel, (E2,), generator=g), torch.randint(0, nrel, (E2,), generator=g),
                                 torch.randint(0, Ne, (E2,), generator=g)], dim=1) if E2 else torch.zeros(0, 4, dtype=torch.long)
            ents = torch.randint(0, Ne, (max(1, Ne // 3),), generator=g)
            ent_emb, rel_emb = torch.randn(Ne, emb, generator=g), torch.randn(nrel, emb, generator=g)
            torch.manual_seed(it)
            m = SpKBGATModified(ent_emb.clone(), rel_emb.clone(), [d1, d1 * h1], [d1, d1 * h1], 0.0, 0.2, [h1, 1], None).to(d).eval()
            oe, orr, mask = m(None, ents.to(d), (edge.to(d), et.to(d)), quads.to(d))
            sg = m.sparse_gat_1
            nrm = torch.nn.functional.normalize(ent_emb, p=2, dim=1)
            re_, rr_, mk = O.spkbgat_forward(nrm.double(), rel_emb.double(), ents, edge, et, quads if E2 else None,
                                             [a.a.detach().cpu().double() for a in sg.attentions], [a.a_2.detach().cpu().double() for a in sg.attentions],
                                             sg.W.detach().cpu().double(), sg.out_att.a.detach().cpu().double(), sg.out_att.a_2.detach().cpu().double(),
                                             m.W_entities.detach().cpu().double(), 0.2)
            chk("kbgat ent", cfg, oe, re_); chk("kbgat rel", cfg, orr, rr_); chk("kbgat mask", cfg, mask, mk, atol=0, rel=0)
            with torch.no_grad():
                be, br, _ = m.batch_test(None, ents.to(d), (edge.to(d), et.to(d)), quads.to(d), m.entity_embeddings)
            chk("kbgat batch_test ent", cfg, be, re_); chk("kbgat batch_test rel", cfg, br, rr_)
        except Exception as ex:
            chk.bad.append(("exception", cfg, repr(ex)[:300], 0))
    graph_mod.clear_graph_cache()
    assert not chk.bad, chk.bad[:5]


def test_fuzz_wide_state_propagation_vs_oracle():
    """16 random cases of the wide-state forms (160 < S <= 512: split pass + 64-channel chunks forward; both backward products as
    batched GEMMs): arbitrary S (multiples of 4), channel counts across the chunk edges, gather widths up to 24, every non-linearity,
    shared / per-batch start states, with and without gradients — against the float64 oracle."""
    from recon_amd.propagation import propagate
    d_ = dev()
    rs = np.random.RandomState(11)
    chk = _Checker()
    for it in range(16):
        g = torch.Generator().manual_seed(2000 + it)
        S = int(rs.choice([164, 176, 200, 256, 260, 320, 384, 388, 448, 512]))
        C = int(rs.choice([1, 7, 63, 64, 65, 128, 150]))
        dd = int(rs.choice([1, 4, 16, 24])); L = int(rs.randint(1, 4)); B = int(rs.choice([1, 3, 9, 20]))
        act = str(rs.choice(["relu", "tanh", "linear"])); per_batch = bool(rs.randint(0, 2)); grad = bool(rs.randint(0, 2))
        cfg = ("wide", it, S, C, dd, L, B, act, per_batch, grad)
        adjs = [(torch.rand(B, S, S, generator=g) - 0.45) * (1.5 / S ** 0.5) for _ in range(L)]
        for a in adjs:
            a[:, rs.randint(0, S)] *= 1e-4                              # a row far below the others: per-row scales
            a[:, :, rs.randint(0, S)] *= 20.0
        h0 = torch.randn(B, C, S, 1, generator=g) if per_batch else torch.randn(C, S, 1, generator=g)
        head = torch.randint(0, S, (C, dd), generator=g); tail = torch.randint(0, S, (C, dd), generator=g)
        Gr = torch.randn(B, C, dd * L, generator=g)

        def run(device, prop, dt):
            A = [a.clone().to(device=device, dtype=dt).requires_grad_(grad) for a in adjs]
            h = h0.clone().to(device=device, dtype=dt).requires_grad_(grad and per_batch)
            out = prop(A, h, act, head.to(device), tail.to(device))
            if grad:
                (out * Gr.to(device=device, dtype=dt)).sum().backward()
            return [out.detach()] + ([a.grad for a in A] + ([h.grad] if per_batch else []) if grad else [])
        got = run(d_, propagate, torch.float32)
        ref = run("cpu", lambda *a: O.propagate(*a, as_gemm=True), torch.float64)
        for k, (x_, y_) in enumerate(zip(got, ref)):
            chk("wide[%d]" % k, cfg, x_, y_, atol=1e-4 if k == 0 else 1e-5, rel=1e-5 if k == 0 else 1e-4)
    assert not chk.bad, chk.bad[:5]


def test_fuzz_wide_state_training_paths_vs_oracle():
    """Ten random GP-GNN problems with 11 .. 32 nodes (float32, block-structured gather indices): forward and EVERY gradient of the dense
    route (block adjacency + propagate: backward chain and d A products on the two-term f16 kernels) and of propagate_blocks (no adjacency
    in either direction) against the float64 oracle.  (A longer run of the same sweep, 24 cases: worst relative error 1.1e-6.)"""
    import random
    from recon_amd.propagation import (propagate, propagate_blocks, build_block_adjacency, get_head_indices, get_tail_indices, make_start_embedding)
    d_ = dev()
    rng = random.Random(7)
    chk = _Checker()
    for case in range(10):
        n, L, B = rng.randint(11, 32), rng.randint(1, 3), rng.choice([1, 2, 3, 5])
        act, per_batch, mode = rng.choice(["relu", "tanh", "linear"]), rng.random() < 0.6, ("dense", "blocks")[case % 2]
        d = 8
        Cn, S, dd = n * (n - 1), 16 * n, 16
        g = torch.Generator().manual_seed(977 * case + n)
        Ts = [torch.relu(torch.randn(B, Cn, dd * dd, generator=g)) * (0.6 / n) for _ in range(L)]
        ident = torch.eye(dd) + 0.02 * torch.randn(dd, dd, generator=g)
        tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
        h0 = (torch.randn(B, Cn, S, 1, generator=g) if per_batch else torch.randn(Cn, S, 1, generator=g)) * tmpl
        head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
        tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
        Gr = torch.randn(B, Cn, dd * L, generator=g)

        def run(device, dt, fn):
            Tl = [t.clone().to(device=device, dtype=dt).requires_grad_(True) for t in Ts]
            I = ident.clone().to(device=device, dtype=dt).requires_grad_(True)
            h = h0.clone().to(device=device, dtype=dt).requires_grad_(True)
            out = fn(Tl, I, h, head.to(device), tail.to(device))
            (out * Gr.to(device=device, dtype=dt)).sum().backward()
            return [out.detach()] + [t.grad for t in Tl] + [I.grad, h.grad]
        ref = run("cpu", torch.float64, lambda Tl, I, h, hd, tl: O.propagate([O.build_block_adjacency(t, I, n) for t in Tl], h, act, hd, tl, as_gemm=True))
        if mode == "dense":
            got = run(d_, torch.float32, lambda Tl, I, h, hd, tl: propagate([build_block_adjacency(t, I, n) for t in Tl], h, act, hd, tl))
        else:
            got = run(d_, torch.float32, lambda Tl, I, h, hd, tl: propagate_blocks(Tl, I, n, h, act, hd, tl))
        cfg = dict(n=n, L=L, B=B, act=act, per_batch=per_batch, mode=mode)
        for i, (a, b) in enumerate(zip(got, ref)):
            chk("tensor %d" % i, cfg, a, b, atol=1e-6, rel=2e-5)
    assert not chk.bad, chk.bad[:4]
