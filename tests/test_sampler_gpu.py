"""GPU tests of the stage-A batch builders (recon_amd/sampler.py): exact integer equality, in order, with the
reference's own outputs (tests/golden/sampler*.npz, produced by running GAT/create_batch.py Corpus) and with the oracle on
larger random knowledge graphs; and the sampler feeding SpKBGATModified end to end."""
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


@pytest.fixture(params=["kernels", "kernels_wave_per_source", "torch_primitives"])
def impl(request, monkeypatch, recon_config):
    """The implementations of the batch builders: csrc/sampler.hip (the default: a workgroup per source in the 2-hop walk where its tables
    fit in LDS), the same with the one-wave-per-source walk forced (RECON_KG_NHOP=w: what large entity sets take), and the torch-primitive assembly."""
    from recon_amd import sampler
    monkeypatch.setattr(sampler, "_KERNELS", request.param != "torch_primitives")
    if request.param == "kernels_wave_per_source":
        recon_config("RECON_KG_NHOP", "w")
    return request.param


@pytest.mark.parametrize("name", ["sampler1_small", "sampler2_medium"])
def test_sampler_golden(name, impl):
    from recon_amd.sampler import KGNeighbourSampler
    g = load_golden(name)
    sm = KGNeighbourSampler(T(g["adj_indices"]).to(dev()), T(g["adj_values"]).to(dev()), int(g["n_ent"]))
    for b in range(int(g["n_batches"])):
        ents = T(g["b%d_entities" % b]).to(dev())
        (edge, et), (ss, ts) = sm.batch_adj_data(ents)
        np.testing.assert_array_equal(edge.cpu().numpy(), g["b%d_edge" % b])
        np.testing.assert_array_equal(et.cpu().numpy(), g["b%d_edge_type" % b])
        assert ss.tolist() == g["b%d_sources" % b].tolist() and ts.tolist() == g["b%d_targets" % b].tolist()
        np.testing.assert_array_equal(sm.batch_nhop_neighbors(ents).cpu().numpy(), g["b%d_nhop" % b])
        np.testing.assert_array_equal(sm.batch_nhop_neighbors(ents, partial_2hop=True).cpu().numpy(), g["b%d_nhop_partial" % b])


@pytest.mark.parametrize("Ne,Tn,n_rel,B,seed", [(500, 6000, 20, 128, 5), (3000, 9000, 237, 128, 6), (40, 2000, 3, 40, 7), (64, 0, 4, 8, 8),
                                                (2000, 60000, 50, 700, 9), (20011, 30000, 11, 256, 10)])   # the last: more entities than one pass of the mark compaction covers
def test_sampler_vs_oracle_random(Ne, Tn, n_rel, B, seed, impl):
    """Dense, sparse, tiny-and-saturated and empty graphs; batch entities in random order (as the reference iterates a shuffled list)."""
    from recon_amd.sampler import KGNeighbourSampler
    rs = np.random.RandomState(seed)
    adj = T(np.stack([rs.randint(0, Ne, Tn), rs.randint(0, max(1, Ne * 3 // 4), Tn)])).long()
    val = T(rs.randint(0, n_rel, Tn)).long()
    sm = KGNeighbourSampler(adj.to(dev()), val.to(dev()), Ne)
    graph = O.kg_graph(adj, val)
    n1, n2 = O.kg_further_neighbors(graph, 1), O.kg_further_neighbors(graph, 2)
    ents = rs.permutation(Ne)[:B].tolist()
    (edge, et), (ss, ts) = sm.batch_adj_data(torch.tensor(ents, device=dev()))
    e2, t2, sset, tset = O.kg_batch_adj_data(n1, ents)
    assert torch.equal(edge.cpu(), e2) and torch.equal(et.cpu(), t2)
    assert ss.tolist() == sorted(sset) and ts.tolist() == sorted(tset)
    q = sm.batch_nhop_neighbors(torch.tensor(ents, device=dev()))
    np.testing.assert_array_equal(q.cpu().numpy(), O.kg_batch_nhop_neighbors(n2, ents))
    if Tn:
        assert edge.shape[1] > 0 and q.shape[0] > 0
    # a second batch through the same sampler: the kernels' marks and counters must have come back zeroed; duplicates in the batch
    ents2 = rs.permutation(Ne)[:max(1, B // 3)].tolist()
    ents2 = ents2 + ents2[:2]
    (edge, et), (ss, ts) = sm.batch_adj_data(torch.tensor(ents2, device=dev()))
    e2, t2, sset, tset = O.kg_batch_adj_data(n1, ents2)
    assert torch.equal(edge.cpu(), e2) and torch.equal(et.cpu(), t2)
    assert ss.tolist() == sorted(sset) and ts.tolist() == sorted(tset)
    np.testing.assert_array_equal(sm.batch_nhop_neighbors(ss, partial_2hop=True).cpu().numpy(), O.kg_batch_nhop_neighbors(n2, ss.tolist(), partial_2hop=True))


def test_sampler_rejects_bad_input():
    from recon_amd.sampler import KGNeighbourSampler
    with pytest.raises(RuntimeError):
        KGNeighbourSampler(torch.zeros(2, 3, dtype=torch.long), torch.zeros(3, dtype=torch.long), 4)       # CPU tensors
    with pytest.raises(IndexError):
        KGNeighbourSampler(torch.tensor([[0, 5], [1, 2]], device=dev()), torch.zeros(2, dtype=torch.long, device=dev()), 4)
    sm = KGNeighbourSampler(torch.tensor([[0, 3], [1, 2]], device=dev()), torch.zeros(2, dtype=torch.long, device=dev()), 4)
    with pytest.raises(IndexError):                                   # an unknown entity in the batch (the reference: KeyError)
        sm.batch_adj_data(torch.tensor([1, 4], device=dev()))
    with pytest.raises(IndexError):
        sm.batch_nhop_neighbors(torch.tensor([-1], device=dev()))


def test_sampler_feeds_spkbgat():
    """One stage-A iteration assembled on the device: sampler -> SpKBGATModified (whole entity table, one entity batch) -> the
    same forward on the edges the ORACLE's builders produce.  Bitwise equal: the edge lists are identical, the kernels deterministic."""
    from recon_amd.sampler import KGNeighbourSampler
    from recon_amd.models import SpKBGATModified
    rs = np.random.RandomState(11)
    Ne, Tn, n_rel, B = 300, 2500, 11, 32
    adj = T(np.stack([rs.randint(0, Ne, Tn), rs.randint(0, Ne, Tn)])).long()
    val = T(rs.randint(0, n_rel, Tn)).long()
    d = dev()
    sm = KGNeighbourSampler(adj.to(d), val.to(d), Ne)
    ents = rs.permutation(Ne)[:B].tolist()
    g = torch.Generator().manual_seed(0)
    ent_emb, rel_emb = torch.randn(Ne, 16, generator=g), torch.randn(n_rel, 16, generator=g)
    torch.manual_seed(0)
    m = SpKBGATModified(ent_emb.clone(), rel_emb.clone(), [8, 16], [16, 16], 0.0, 0.2, [2, 2], None).to(d).eval()
    ents_d = torch.tensor(ents, device=d)
    (edge, et), (ss, ts) = sm.batch_adj_data(ents_d)
    nhop = sm.batch_nhop_neighbors(ents_d)
    batch_all = torch.unique(torch.cat((ss, ts)))
    with torch.no_grad():
        out_e, out_r, mask = m(None, batch_all, (edge, et), nhop)
    graph = O.kg_graph(adj, val)
    e2, t2, sset, tset = O.kg_batch_adj_data(O.kg_further_neighbors(graph, 1), ents)
    q2 = torch.from_numpy(O.kg_batch_nhop_neighbors(O.kg_further_neighbors(graph, 2), ents).astype(np.int64))
    torch.manual_seed(0)
    m2 = SpKBGATModified(ent_emb.clone(), rel_emb.clone(), [8, 16], [16, 16], 0.0, 0.2, [2, 2], None).to(d).eval()
    with torch.no_grad():
        out_e2, out_r2, mask2 = m2(None, torch.tensor(sorted(sset | tset), device=d), (e2.to(d), t2.to(d)), q2.to(d))
    assert torch.equal(out_e, out_e2) and torch.equal(out_r, out_r2) and torch.equal(mask, mask2)
    assert torch.isfinite(out_e).all() and float(mask.sum()) == len(sset | tset)


@pytest.mark.parametrize("Ne,E1,E2,B,seed", [(500, 3000, 9000, 40, 0), (14541, 2400, 27000, 128, 1), (60, 80, 0, 25, 2), (300, 1, 5, 3, 3), (100, 700, 300, 0, 4)])
def test_prune_edges_vs_numpy(Ne, E1, E2, B, seed):
    """sampler.prune_edges: the edges whose destination lies in mask U {src(e) : mask[dst(e)]}, in input order, with their types and
    positions — exact integer equality with the set arithmetic done in numpy."""
    from recon_amd.sampler import prune_edges
    rs = np.random.RandomState(seed)
    d = dev()
    mask = np.zeros(Ne, dtype=np.float32)
    mask[rs.permutation(Ne)[:B]] = 1.0
    e1, t1 = rs.randint(0, Ne, size=(2, E1)), rs.randint(0, 7, size=E1)
    e2, t2 = rs.randint(0, Ne, size=(2, E2)), rs.randint(0, 7, size=(E2, 2))
    need = mask != 0
    for e in (e1, e2):
        need[e[1][mask[e[0]] != 0]] = True
    k1, k2 = need[e1[0]], need[e2[0]]
    nh = (T(e2).to(d), T(t2).to(d)) if E2 else (torch.tensor([]), torch.tensor([]))
    oe, ot, on, ont, pos = prune_edges(T(mask).to(d), T(e1).to(d), T(t1).to(d), nh[0], nh[1], want_pos=True)
    np.testing.assert_array_equal(oe.cpu().numpy(), e1[:, k1])
    np.testing.assert_array_equal(ot.cpu().numpy(), t1[k1])
    assert ot.is_contiguous()
    if k2.any():
        np.testing.assert_array_equal(on.cpu().numpy(), e2[:, k2])
        np.testing.assert_array_equal(ont.cpu().numpy(), t2[k2])
        assert ont.is_contiguous()
    else:
        assert on.numel() == 0 and ont.numel() == 0
    np.testing.assert_array_equal(pos.cpu().numpy(), np.concatenate([np.nonzero(k1)[0], E1 + np.nonzero(k2)[0]]))
    # the whole-call form: mask made from the batch's ids, n-hop list read from quadruples, survivors of both lists side by side in one buffer
    from recon_amd.sampler import prune_batch
    from recon_amd.graph import prepare_graph, clear_graph_cache
    be = np.nonzero(mask)[0][rs.permutation(int(mask.sum()))]
    be = np.concatenate([be, be[:3]])                                    # duplicates: the reference takes torch.unique
    quads = np.stack([e2[1], t2[:, 0], t2[:, 1], e2[0]], axis=1) if E2 else np.zeros((0, 4), dtype=np.int64)
    m2, oe2, ot2, on2, ont2, pos2 = prune_batch(T(be).to(d), T(e1).to(d), T(t1).to(d), T(quads).to(d) if E2 else torch.tensor([], dtype=torch.long), Ne, want_pos=True,
                                                table_rows=7)
    if k2.any():                                                         # the table row of every surviving edge, handed to the layers' extended-index cache
        from recon_amd.gat_layers import _extended_index
        ext = _extended_index(ot2, 7, int(k2.sum()))
        np.testing.assert_array_equal(ext.cpu().numpy(), np.concatenate([t1[k1], 7 + np.arange(int(k2.sum()))]))
    np.testing.assert_array_equal(m2.cpu().numpy(), mask)
    for a_, b_ in ((oe2, oe), (ot2, ot), (pos2, pos)):
        assert torch.equal(a_, b_)
    if k2.any():
        assert torch.equal(on2, on) and torch.equal(ont2, ont)
        clear_graph_cache()
        g_join = prepare_graph(oe2, on2, Ne)                             # takes the joined view: no cat
        assert g_join.edge.data_ptr() == oe2.data_ptr()
        clear_graph_cache()
        g_cat = prepare_graph(oe.contiguous(), on.contiguous(), Ne)
        for nm in ("rowptr_dst", "eid", "src", "dst", "rowptr_src", "slot_by_src"):
            assert torch.equal(getattr(g_join, nm), getattr(g_cat, nm)), nm
        clear_graph_cache()
    else:
        assert on2.numel() == 0
    if B:
        with pytest.raises(IndexError):
            prune_batch(torch.tensor([0, Ne], device=d), T(e1).to(d), T(t1).to(d), torch.tensor([], dtype=torch.long), Ne)


@pytest.mark.parametrize("drop", [0.0, 0.3])
def test_spkbgat_with_and_without_dead_row_pruning(drop, monkeypatch):
    """SpKBGATModified on a sampled batch with the edges into discarded rows dropped up front (models.PRUNE_DEAD_ROWS) against the same
    model evaluating every row as the reference does: outputs and every parameter's gradient (train mode: the same per-edge factors,
    handed to the pruned run by position)."""
    from recon_amd import models
    from recon_amd.sampler import KGNeighbourSampler
    rs = np.random.RandomState(3)
    Ne, Tn, n_rel, B = 2000, 30000, 11, 64
    p = 1.0 / np.arange(1, Ne + 1) ** 0.8
    p /= p.sum()
    adj = T(np.stack([rs.choice(Ne, size=Tn, p=p), rs.choice(Ne, size=Tn, p=p)])).long()
    val = T(rs.randint(0, n_rel, Tn)).long()
    d = dev()
    sm = KGNeighbourSampler(adj.to(d), val.to(d), Ne)
    ents = torch.tensor(rs.permutation(Ne)[:B].tolist(), device=d)
    (edge, et), (ss, ts) = sm.batch_adj_data(ents)
    nhop = sm.batch_nhop_neighbors(ss)
    E = edge.shape[1] + nhop.shape[0]
    g = torch.Generator().manual_seed(0)
    ent_emb, rel_emb = torch.randn(Ne, 16, generator=g), torch.randn(n_rel, 16, generator=g)
    G = torch.randn(Ne, 16, generator=g).to(d)
    keeps = [((torch.rand(E, generator=g) > drop).float() / (1.0 - drop)).to(d) for _ in range(3)]
    layer_keep = ((torch.rand(Ne, 16, generator=g) > drop).float() / (1.0 - drop)).to(d)
    res = {}
    for prune in (True, False):
        monkeypatch.setattr(models, "PRUNE_DEAD_ROWS", prune)
        monkeypatch.setattr(models, "KEEP_PRUNED_POSITIONS", True)
        torch.manual_seed(0)
        m = models.SpKBGATModified(ent_emb.clone(), rel_emb.clone(), [8, 16], [16, 16], drop, 0.2, [2, 2], None).to(d)
        m.train(drop > 0)
        if drop > 0:
            sg = m.sparse_gat_1
            kept = lambda mk: mk if m._pruned_pos is None else mk[m._pruned_pos]
            for h, att in enumerate(list(sg.attentions) + [sg.out_att]):
                att.draw_keep = (lambda mk: (lambda E_, device: kept(mk).view(1, E_)))(keeps[h])
            sg.dropout_layer.forward = lambda x: x * layer_keep
        out_e, out_r, mask = m(None, ss, (edge, et), nhop)
        ((out_e * G).sum() + out_r.sum()).backward()
        res[prune] = [out_e.detach(), out_r.detach()] + [p_.grad for _, p_ in sorted(m.named_parameters()) if p_.grad is not None]
        if prune:
            assert m._pruned_pos.numel() < E // 4                      # the batch's neighbours are not batch entities: most edges go
    assert len(res[True]) == len(res[False]) > 4
    for a_, b_ in zip(res[True], res[False]):
        tol = 1e-5 + 1e-4 * float(b_.abs().max())
        assert float((a_ - b_).abs().max()) <= tol


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["loss1_small", "loss2_wide", "loss3_ratio1"])
def test_batch_gat_loss_golden(name, golden):
    """recon_amd.losses.batch_gat_loss (one launch forward; gradient rows + two fixed-order segment sums backward) against the reference's
    own batch_gat_loss on the same triples: loss and both table gradients; and against the fallback (the reference's op sequence on
    gather_rows), which any other loss function takes."""
    from recon_amd.losses import batch_gat_loss
    g = golden(name)
    d = torch.device("cuda:0")
    tri = torch.from_numpy(g["train_indices"]).to(d)
    margin, ratio = float(g["margin"]), int(g["ratio"])
    for loss_fn in (torch.nn.MarginRankingLoss(margin=margin), lambda a, b, y: torch.nn.functional.margin_ranking_loss(a, b, y, margin=margin)):
        ent = torch.from_numpy(g["entity"]).to(d).requires_grad_(True)
        rel = torch.from_numpy(g["relation"]).to(d).requires_grad_(True)
        loss = batch_gat_loss(loss_fn, tri, ent, rel, valid_invalid_ratio_gat=ratio)
        (loss * 1.5).backward()
        np.testing.assert_allclose(loss.detach().cpu().numpy(), g["loss"], rtol=1e-5, atol=1e-6)
        np.testing.assert_allclose(ent.grad.cpu().numpy(), 1.5 * g["g_entity"], rtol=1e-5, atol=1e-7)
        np.testing.assert_allclose(rel.grad.cpu().numpy(), 1.5 * g["g_relation"], rtol=1e-5, atol=1e-7)


@pytest.mark.gpu
def test_batch_gat_loss_vs_oracle_at_stage_a_size_and_odd_width():
    """Stage-A sizes (2 000 positives, 200-wide tables, ratio 2) and a width that is not a multiple of four (the scalar path); twice in a
    row (the arrival counter must come back to zero); ids out of range raise like the reference's indexing."""
    from recon_amd.losses import batch_gat_loss
    from oracle import recon_oracle as O
    d = torch.device("cuda:0")
    for n_ent, n_rel, D, n_pos, ratio in ((14541, 237, 200, 2000, 2), (50, 7, 37, 101, 3)):
        gen = torch.Generator().manual_seed(D)
        ent, rel = torch.randn(n_ent, D, generator=gen), torch.randn(n_rel, D, generator=gen)
        pos = torch.stack((torch.randint(0, n_ent, (n_pos,), generator=gen), torch.randint(0, n_rel, (n_pos,), generator=gen),
                           torch.randint(0, n_ent, (n_pos,), generator=gen)), 1)
        neg = pos.repeat(2 * ratio, 1)
        neg[: neg.shape[0] // 2, 0] = torch.randint(0, n_ent, (neg.shape[0] // 2,), generator=gen)
        neg[neg.shape[0] // 2:, 2] = torch.randint(0, n_ent, (neg.shape[0] - neg.shape[0] // 2,), generator=gen)
        tri = torch.cat((pos, neg))
        er, rr = ent.clone().double().requires_grad_(True), rel.clone().double().requires_grad_(True)
        ref = O.batch_gat_loss(tri, er, rr, ratio, margin=2.0)
        ref.backward()
        for _ in range(2):
            ed, rd = ent.to(d).requires_grad_(True), rel.to(d).requires_grad_(True)
            loss = batch_gat_loss(torch.nn.MarginRankingLoss(margin=2.0), tri.to(d), ed, rd, valid_invalid_ratio_gat=ratio)
            loss.backward()
            np.testing.assert_allclose(loss.item(), ref.item(), rtol=2e-5)
            np.testing.assert_allclose(ed.grad.cpu().numpy(), er.grad.float().numpy(), rtol=1e-4, atol=1e-6)
            np.testing.assert_allclose(rd.grad.cpu().numpy(), rr.grad.float().numpy(), rtol=1e-4, atol=1e-6)
    bad = tri.clone()
    bad[3, 2] = n_ent
    with pytest.raises(IndexError):
        batch_gat_loss(torch.nn.MarginRankingLoss(margin=1.0), bad.to(d), ent.to(d), rel.to(d), valid_invalid_ratio_gat=ratio)


@pytest.mark.gpu
def test_batch_gat_loss_propagates_nan_and_takes_indices_from_the_host():
    """(1) a NaN embedding row must reach the loss (the reference's clamp_min propagates NaN and GAT/main.py:374 asserts on it;
    `fmaxf(0, NaN)` is 0) — fused path and fallback — and raise the device NaN word; (2) the reference indexes its CUDA tables with whatever
    LongTensor it is given, a CPU one included: same loss as with device indices, no device fault."""
    from recon_amd.losses import batch_gat_loss
    from recon_amd.gat_layers import nan_raised, enable_nan_flag
    d = torch.device("cuda:0")
    enable_nan_flag(d)
    gen = torch.Generator().manual_seed(5)
    n_ent, n_rel, D, n_pos, ratio = 40, 5, 16, 12, 2
    ent, rel = torch.randn(n_ent, D, generator=gen), torch.randn(n_rel, D, generator=gen)
    pos = torch.stack((torch.randint(0, n_ent, (n_pos,), generator=gen), torch.randint(0, n_rel, (n_pos,), generator=gen),
                       torch.randint(0, n_ent, (n_pos,), generator=gen)), 1)
    neg = pos.repeat(2 * ratio, 1)
    neg[:, 0] = torch.randint(0, n_ent, (neg.shape[0],), generator=gen)
    tri = torch.cat((pos, neg))
    fn = torch.nn.MarginRankingLoss(margin=1.0)
    assert not nan_raised(d)
    clean = batch_gat_loss(fn, tri.to(d), ent.to(d), rel.to(d), valid_invalid_ratio_gat=ratio)
    assert torch.isfinite(clean) and not nan_raised(d)
    host_idx = batch_gat_loss(fn, tri, ent.to(d), rel.to(d), valid_invalid_ratio_gat=ratio)               # CPU LongTensor into CUDA tables
    assert torch.equal(host_idx, clean)
    bad = ent.clone()
    bad[int(pos[3, 0])] = float("nan")
    for f in (fn, lambda a, b, y: torch.nn.functional.margin_ranking_loss(a, b, y, margin=1.0)):          # fused, then the fallback
        e = bad.to(d).requires_grad_(True)
        loss = batch_gat_loss(f, tri.to(d), e, rel.to(d), valid_invalid_ratio_gat=ratio)
        assert torch.isnan(loss), "a NaN row must reach the loss"
    assert nan_raised(d), "the fused loss raises the device word"
    big = ent.clone() * 3e37                                                                               # two overflowing L1 norms: inf - inf
    loss = batch_gat_loss(fn, tri.to(d), big.to(d), rel.to(d), valid_invalid_ratio_gat=ratio)
    assert torch.isnan(loss) and nan_raised(d)
