"""GPU parity tests of the GP-GNN propagation step in bfloat16 (BASELINE.json configs[2] / configs[4]; csrc/prop_b16.hip): bf16
storage, fp32 accumulation, every state rounded once per hop.

Method (the one tests/test_prop_gpu.py::test_gcn_bf16_vs_oracle uses): the forward against the fp32 oracle on the SAME bf16-rounded
operands with the oracle rounding its states hop by hop (`storage=torch.bfloat16`) — what remains is the order of the fp32 sums,
i.e. an occasional last-place difference of a bf16 value (2^-8 relative) carried through the hops; the gradients against the
oracle's closed-form backward (pinned to its autograd by tests/test_oracle_golden.py) evaluated from the states the bf16 forward
itself saved, so that the ReLU mask is the forward's own.  Reference math: models/models.py:240-274."""
import numpy as np
import pytest
import torch

from oracle import recon_oracle as O
from test_gat_gpu import close, dev

pytestmark = pytest.mark.gpu
BF = torch.bfloat16


def _bf(t):
    return t.to(BF)


def _check_b16(out, grads, adjs_b, h0_b, head, tail, act, Gr_b, states, per_batch, what, fwd_rel=1.5e-2, bwd_rel=2e-2):
    """out / grads (g_adj list, g_h0) of the kernels vs the oracle on the bf16 operands; `states` = what the forward saved."""
    adjs_f = [a.float().cpu() for a in adjs_b]
    h0_f = h0_b.float().cpu()
    ref, ref_states = O.propagate(adjs_f, h0_f, act, head, tail, as_gemm=True, storage=BF, return_states=True)
    assert out.dtype == BF
    close(out.float(), ref, atol=1e-3, rel_to_max=fwd_rel, what=what + " out")
    if grads is None:
        return
    L = len(adjs_b)
    st = [states[l].float().cpu() for l in range(L)]
    for l in range(L):
        close(st[l], ref_states[l], atol=1e-3, rel_to_max=fwd_rel, what=what + " state %d" % (l + 1))
    g_adj_r, g_h_r = O.propagate_backward(adjs_f, h0_f, st, act, head, tail, Gr_b.float().cpu(), storage=BF)
    g_adj, g_h0 = grads
    for l in range(L):
        assert g_adj[l].dtype == BF
        close(g_adj[l].float(), g_adj_r[l], atol=1e-3, rel_to_max=bwd_rel, what=what + " g_adj[%d]" % l)
    if g_h0 is not None:
        ref_h = g_h_r if per_batch else g_h_r.sum(0)
        close(g_h0.float().reshape(ref_h.shape), ref_h, atol=1e-3, rel_to_max=bwd_rel, what=what + " g_h0")


def _run_b16(adjs_b, h0_b, head, tail, act, Gr_b, grad=True):
    from recon_amd import propagation as P
    d_ = dev()
    A = [a.clone().to(d_).requires_grad_(grad) for a in adjs_b]
    h = h0_b.clone().to(d_).requires_grad_(grad)
    out, states = P.propagate(A, h, act, head.to(d_), tail.to(d_), return_states=True)
    if not grad:
        return out.detach(), None, None
    (out.float() * Gr_b.to(d_).float()).sum().backward()
    return out.detach(), ([a.grad for a in A], h.grad), states


@pytest.mark.parametrize("n,d,L,B,act,per_batch", [
    (9, 8, 3, 5, "relu", True),        # model_params.json sizes: S = 144, C = 72 (NKS = 5, NTC = 5, half a K step out of range)
    (9, 8, 3, 600, "relu", False),     # more graphs than workgroups: the prefetch crosses graph boundaries; shared h0
    (4, 8, 2, 7, "tanh", True),        # S = 64, C = 12
    (10, 8, 3, 3, "linear", True),     # S = 160, C = 90: the largest fused shape
    (2, 8, 4, 9, "relu", False),       # S = 32, C = 2: one K step, two waves; four hops (even: the images swap roles)
    (6, 4, 1, 4, "relu", True),        # 2d = 8: S = 48, gather width 8
    (3, 24, 2, 3, "tanh", True),       # 2d = 48: S = 144, C = 6, 288 gather items
])
@pytest.mark.parametrize("form", ["fused", "gemm"])
def test_propagation_b16_vs_oracle(n, d, L, B, act, per_batch, form, recon_config):
    """Small states (S <= 160, C <= 96): all hops of a graph in one workgroup (`fused`), and the same problems through the batched-GEMM
    form the wide states use (`gemm`: RECON_PROP_B16=g) — forward, saved states and every gradient."""
    if form == "gemm":
        recon_config("RECON_PROP_B16", "g")
    from recon_amd.propagation import make_start_embedding, get_head_indices, get_tail_indices
    C, S, dd = n * (n - 1), 2 * d * n, 2 * d
    g = torch.Generator().manual_seed(n * 100 + d)
    adjs = [_bf((torch.rand(B, S, S, generator=g) - 0.42) * (2.4 / S ** 0.5)) for _ in range(L)]
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = _bf((torch.randn(B, C, S, 1, generator=g) * tmpl) if per_batch else tmpl)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0])
    Gr = _bf(torch.randn(B, C, dd * L, generator=g))
    out, grads, states = _run_b16(adjs, h0, head, tail, act, Gr)
    _check_b16(out, grads, adjs, h0, head, tail, act, Gr, states, per_batch, "b16 %s n=%d" % (form, n))
    out_i, _, _ = _run_b16(adjs, h0, head, tail, act, Gr, grad=False)     # inference: no saved states in the fused form
    assert torch.equal(out_i, out)


@pytest.mark.parametrize("S,C,dd,L,B,act,per_batch", [
    (512, 130, 16, 3, 3, "relu", True),      # wide states: M = 130 = one full tile + 2 rows; two channel chunks, the second of 2 channels
    (176, 5, 8, 3, 4, "linear", True),       # K = 176: a partial stage (5.5 K steps), M = 5
    (168, 20, 4, 2, 9, "relu", False),       # S % 16 != 0 (S % 8 == 0): partial row / column tiles; more than 8 graphs: two XCD rounds
    (256, 96, 6, 2, 11, "tanh", True),       # exactly two tiles each way; NKS = 8, two row tiles per wave
    (144, 150, 16, 2, 3, "relu", True),      # S <= 160 but C > 96: the mid-size tile (144 x 144) with two row tiles
    (40, 3, 2, 8, 2, "tanh", False),         # eight hops, tiny states, duplicates in the gather indices
    (192, 40, 8, 3, 9, "relu", True),        # NKS = 6: twelve row tiles on eight waves (two waves idle)
    (320, 200, 16, 2, 3, "relu", False),     # NKS = 10, three row tiles per wave (an unpaired one in the epilogue), shared h0
    (384, 129, 4, 2, 2, "tanh", True),       # NKS = 12
    (448, 70, 16, 2, 2, "linear", True),     # NKS = 14: 28 row tiles on 8 x 4
    (512, 300, 20, 2, 2, "relu", True),      # gather width 20: more items than the position table holds
])
@pytest.mark.parametrize("form", ["auto", "gemm"])
def test_propagation_b16_gemm_form_vs_oracle(S, C, dd, L, B, act, per_batch, form, recon_config):
    """Shapes the small fused kernel does not take.  `auto`: S % 64 == 0, 192 <= S <= 512 runs the wide fused kernel (the state of 128
    channels resident in LDS for all hops), everything else one batched GEMM per hop over the graphs; `gemm`: the GEMM form everywhere.
    Arbitrary adjacencies, start states and gather indices (duplicates included); the backward is the GEMM form in both."""
    if form == "gemm":
        recon_config("RECON_PROP_B16", "g")
    g = torch.Generator().manual_seed(S + C)
    adjs = [_bf((torch.rand(B, S, S, generator=g) - 0.45) * (2.0 / S ** 0.5)) for _ in range(L)]
    h0 = _bf(torch.randn(B, C, S, 1, generator=g) if per_batch else torch.randn(C, S, 1, generator=g))
    head = torch.randint(0, S, (C, dd), generator=g)
    tail = torch.randint(0, S, (C, dd), generator=g)
    Gr = _bf(torch.randn(B, C, dd * L, generator=g))
    out, grads, states = _run_b16(adjs, h0, head, tail, act, Gr)
    _check_b16(out, grads, adjs, h0, head, tail, act, Gr, states, per_batch, "b16 gemm S=%d" % S)


def test_propagation_b16_unaligned_state_size_runs_on_the_float32_kernels():
    """S % 8 != 0 (n = 5, d = 3: S = 30): bf16 tensors around the float32 kernels — same values as the float32 call, rounded once."""
    from recon_amd.propagation import propagate
    d_ = dev()
    g = torch.Generator().manual_seed(3)
    B, C, S, dd, L = 4, 20, 30, 6, 2
    adjs = [_bf((torch.rand(B, S, S, generator=g) - 0.4) * 0.4) for _ in range(L)]
    h0 = _bf(torch.randn(B, C, S, 1, generator=g))
    head, tail = torch.randint(0, S, (C, dd), generator=g), torch.randint(0, S, (C, dd), generator=g)
    A = [a.to(d_).requires_grad_(True) for a in adjs]
    out = propagate(A, h0.to(d_), "relu", head.to(d_), tail.to(d_))
    assert out.dtype == BF
    ref = propagate([a.float().to(d_) for a in adjs], h0.float().to(d_), "relu", head.to(d_), tail.to(d_))
    assert torch.equal(out, ref.to(BF))
    out.float().sum().backward()
    assert A[0].grad is not None and A[0].grad.dtype == BF


@pytest.mark.parametrize("n,B", [(9, 6), (4, 3), (2, 5), (10, 2)])
def test_block_adjacency_b16(n, B):
    """P1 on bf16 tensors: a permutation — bit-equal to the oracle's; the backward un-permutes (exact) and sums the diagonal blocks
    (fp32 sum, rounded once)."""
    from recon_amd.propagation import build_block_adjacency
    d_ = dev()
    dd, C = 16, n * (n - 1)
    g = torch.Generator().manual_seed(n)
    T = _bf(torch.randn(B, C, dd * dd, generator=g))
    I = _bf(torch.eye(dd) + 0.1 * torch.randn(dd, dd, generator=g))
    Td, Id = T.to(d_).requires_grad_(True), I.to(d_).requires_grad_(True)
    A = build_block_adjacency(Td, Id, n)
    ref = O.build_block_adjacency(T.float(), I.float(), n)
    assert A.dtype == BF and torch.equal(A.float().cpu(), ref)
    Gr = _bf(torch.randn(B, n * dd, n * dd, generator=g))
    A.backward(Gr.to(d_))
    Tr, Ir = T.float().requires_grad_(True), I.float().requires_grad_(True)
    O.build_block_adjacency(Tr, Ir, n).backward(Gr.float())
    assert torch.equal(Td.grad.float().cpu(), Tr.grad)
    close(Id.grad.float(), Ir.grad, atol=1e-3, rel_to_max=8e-3, what="g_identity bf16")


@pytest.mark.parametrize("n,L,B,act,per_batch", [(9, 3, 7, "relu", True), (9, 3, 300, "relu", False), (4, 2, 5, "tanh", True), (2, 3, 3, "relu", True),
                                                  (10, 2, 4, "relu", True), (7, 3, 5, "linear", False),
                                                  (12, 2, 3, "relu", True), (16, 3, 9, "tanh", True), (20, 2, 2, "relu", False), (28, 2, 2, "relu", True),
                                                  (32, 3, 3, "relu", True), (11, 2, 3, "relu", True)])      # n = 11: no fused form, materialised
def test_propagate_blocks_b16_inference_matches_materialised(n, L, B, act, per_batch):
    """Block mode of the fused bf16 kernel (the transition tensors read in place, A_l never written) gives the bits of the path through
    build_block_adjacency: the same fragments reach the same MFMAs in the same order."""
    from recon_amd.propagation import build_block_adjacency, propagate, propagate_blocks, make_start_embedding, get_head_indices, get_tail_indices
    d_ = dev()
    d = 8
    dd, C, S = 16, n * (n - 1), 16 * n
    g = torch.Generator().manual_seed(n + L)
    Ts = [_bf(torch.relu(torch.randn(B, C, dd * dd, generator=g)) * (1.5 / S ** 0.5)).to(d_) for _ in range(L)]
    I = _bf(torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g)).to(d_)
    tmpl = torch.from_numpy(make_start_embedding(n, d)).float()
    h0 = _bf((torch.randn(B, C, S, 1, generator=g) * tmpl) if per_batch else tmpl).to(d_)
    head = torch.from_numpy(get_head_indices(n, d, bs=1)[0]).to(d_)
    tail = torch.from_numpy(get_tail_indices(n, d, bs=1)[0]).to(d_)
    with torch.no_grad():
        fused = propagate_blocks(Ts, I, n, h0, act, head, tail)
        plain = propagate([build_block_adjacency(t, I, n) for t in Ts], h0, act, head, tail)
    assert fused.dtype == BF and torch.equal(fused, plain)
    ref = O.propagate([O.build_block_adjacency(t.float().cpu(), I.float().cpu(), n) for t in Ts], h0.float().cpu(), act, head.cpu(), tail.cpu(),
                      as_gemm=True, storage=BF)
    close(fused.float(), ref, atol=1e-3, rel_to_max=1.5e-2, what="blocks b16 n=%d" % n)


@pytest.mark.parametrize("n,L,B,ypost", [(9, 3, 6, "fused"), (9, 3, 6, "kernel"), (11, 2, 3, "fused"), (12, 2, 10, "fused"), (4, 3, 5, "fused")])
def test_propagate_blocks_b16_training_vs_oracle(n, L, B, ypost, recon_config):
    """models/models.py:240-274 on bf16 tensors with gradients, no adjacency materialised in either direction: transition tensors -> relu ->
    [block adjacency read in place] -> L hops -> loss, all gradients (d T_l in T's layout, d identity, d h0) against the oracle's closed
    form run from the forward's own states.  n = 9 / 4: fused small forward; n = 11: every hop a batched GEMM reading T in place; n = 12:
    the wide fused forward.  `kernel`: Y_l by the separate pass instead of the GEMM epilogue (RECON_PROP_B16_YPOST=k)."""
    if ypost == "kernel":
        recon_config("RECON_PROP_B16_YPOST", "k")
    from recon_amd import propagation as P
    d_ = dev()
    d = 8
    dd, C, S = 16, n * (n - 1), 16 * n
    g = torch.Generator().manual_seed(11)
    Ts = [_bf((torch.rand(B, C, dd * dd, generator=g) - 0.3) * (1.5 / S ** 0.5)) for _ in range(L)]
    I = _bf(torch.eye(dd) + 0.05 * torch.randn(dd, dd, generator=g))
    tmpl = torch.from_numpy(P.make_start_embedding(n, d)).float()
    h0 = _bf(torch.randn(B, C, S, 1, generator=g) * tmpl)
    head = torch.from_numpy(P.get_head_indices(n, d, bs=1)[0])
    tail = torch.from_numpy(P.get_tail_indices(n, d, bs=1)[0])
    Gr = _bf(torch.randn(B, C, dd * L, generator=g))
    Td = [t.to(d_).requires_grad_(True) for t in Ts]
    Id, hd = I.to(d_).requires_grad_(True), h0.to(d_).requires_grad_(True)
    out, states = P.propagate_blocks([torch.relu(t) for t in Td], Id, n, hd, "relu", head.to(d_), tail.to(d_), return_states=True)
    (out.float() * Gr.to(d_).float()).sum().backward()
    adjs = [O.build_block_adjacency(torch.relu(t.float()), I.float(), n) for t in Ts]
    ref = O.propagate(adjs, h0.float(), "relu", head, tail, as_gemm=True, storage=BF)
    close(out.float(), ref, atol=1e-3, rel_to_max=1.5e-2, what="blocks b16 training out")
    g_adj, g_h = O.propagate_backward(adjs, h0.float(), [s.float().cpu() for s in states], "relu", head, tail, Gr.float(), storage=BF)
    close(hd.grad.float().reshape(g_h.shape), g_h, atol=1e-3, rel_to_max=2e-2, what="blocks b16 g_h0")
    gI = torch.zeros(dd, dd)
    for l in range(L):
        blocks = g_adj[l].reshape(B, n, dd, n, dd).permute(0, 1, 3, 2, 4)         # [B, i, j, r, c]
        off = torch.stack([blocks[:, i, j] for i in range(n) for j in range(n) if i != j], 1).reshape(B, C, dd * dd)
        gT = off * (Ts[l].float() > 0)
        close(Td[l].grad.float(), gT, atol=1e-3, rel_to_max=2e-2, what="blocks b16 g_T[%d]" % l)
        gI += torch.stack([blocks[:, i, i] for i in range(n)], 1).sum((0, 1))
    close(Id.grad.float(), gI, atol=1e-2, rel_to_max=3e-2, what="blocks b16 g_identity")      # L sums of bf16-rounded per-hop sums


@pytest.mark.parametrize("name", ["gpgnn1_untied", "gpgnn2_tied_n9"])
def test_gpgnn_model_runs_in_bfloat16(name):
    """`GPGNN(...).to(torch.bfloat16)` (models/models.py:85-277 with bfloat16 parameters): the stock encoder in bf16, transition tensors ->
    block adjacency -> propagation through the bfloat16 kernels (block mode, with gradients).  Logits against the reference's own float32
    logits of the same checkpoint (bf16 rounding of every stage: 5e-2 of the largest logit), every trainable parameter gets a finite
    gradient whose direction agrees with the reference's."""
    from conftest import load_golden
    from recon_amd.gpgnn import GPGNN
    g = load_golden(name)
    p = {"max_num_nodes": int(g["n"]), "embedding_dim": int(g["d"]), "layer_number": int(g["L"]), "projection_style": str(g["style"]),
         "non-linear1": "relu", "non-linear": "tanh", "dropout1": 0.0, "position_emb": 3, "units1": 4, "rnn1_layers": 1,
         "bidirectional": 1, "batch_size": int(g["B"])}
    m = GPGNN(p, g["emb"], max_sent_len=4, n_out=3)
    m.load_state_dict({k[3:]: torch.from_numpy(np.asarray(v)) for k, v in g.items() if k.startswith("sd.")}, strict=False)
    m.train().to(dev()).to(BF)
    out = m(torch.from_numpy(g["sent"]).to(dev()), torch.from_numpy(g["mark"]).to(dev()), None)
    assert out.dtype == BF
    close(out.float(), g["out"], atol=2e-3, rel_to_max=5e-2, what="gpgnn bf16 logits")
    (out.float() * torch.from_numpy(g["G"]).to(dev())).sum().backward()
    for k, v in m.named_parameters():
        if "g." + k in g and v.requires_grad:
            assert v.grad is not None and v.grad.dtype == BF and torch.isfinite(v.grad.float()).all(), k
            ref = torch.from_numpy(np.asarray(g["g." + k])).flatten()
            got = v.grad.float().cpu().flatten()
            if ref.norm() > 1e-6:
                cos = float((ref @ got) / (ref.norm() * got.norm() + 1e-30))
                assert cos > 0.98, "grad %s: cosine %.4f against the float32 reference" % (k, cos)
