"""Drop-in replacement for the reference module `GAT/layers.py` (and its byte-identical twin
`GAT_sep_space/layers.py`): same module-level names, same constructor / forward signatures, same
`state_dict` keys (`a`, `a_2`), but the arithmetic runs in hand-written gfx950 kernels behind the
C ABI of include/recon_hip.h.

    from recon_amd.gat_layers import SpGraphAttentionLayer, ConvKB      # GAT/models.py:6

Exports (GAT/layers.py): CUDA, ConvKB, SpecialSpmmFunctionFinal, SpecialSpmmFinal,
SpGraphAttentionLayer.  Extra: `gat_heads(...)`, the fused H-head entry used by recon_amd.models.SpGAT.

No CPU path: tensors must be on an MI355X; a missing librecon_hip.so raises RuntimeError.
"""
import ctypes as C
import os

from collections import OrderedDict

import torch
import torch.nn as nn

from . import _lib
from .graph import prepare_graph, _has_nhop, trust, trusted, trust_bounds

CUDA = torch.cuda.is_available()          # GAT/layers.py:9
_DEBUG_NAN = os.environ.get("RECON_DEBUG_NAN", "0") == "1"
# "auto": aggregate-then-project kernels (csrc/gat_atp.hip) where instantiated, else project-then-aggregate
_GAT_PATH = os.environ.get("RECON_GAT_PATH", "auto")
# RECON_OVERLAP=1 runs the MFMA-bound weight-gradient GEMM of the backward on a side stream next to the HBM-bound
# edge chain.  Off by default: measured neutral on MI355X (0.927 vs 0.922 ms/step at cfg 2) because the GEMM's
# 4 waves/SIMD x 128 registers leave no register file for co-resident edge-kernel waves.
_OVERLAP = os.environ.get("RECON_OVERLAP", "0") == "1"
# Data parallelism (recon_amd/dist.py::OverlappedWeightGradSync): an object with all_reduce_mean(tensor, async_op) -> handle-or-None.  While one
# is installed, the backward of the aggregate-then-project heads reduces its weight gradients across ranks ITSELF and returns them already
# averaged: G = V^T g_h is summed over split-K and sent off before the edge chain (INPUTS) runs, so the big collective travels under it.
_WEIGHT_GRAD_SYNC = None
_GRAD_DEST = {}             # data_ptr of a fused [H, D, W] weight -> (g_a, g_a_2) tensors the heads' backward writes its weight gradients into


class _GradDest:
    """Where one fused weight's gradients may be written, and whether a backward has done so since the owner last released it."""
    __slots__ = ("g_a", "g_a_2", "claimed")

    def __init__(self, g_a, g_a_2):
        self.g_a, self.g_a_2, self.claimed = g_a, g_a_2, False


def set_weight_grad_destination(a, g_a, g_a_2, release=True):
    """Have the backward of gat_heads(..., a, ...) write the gradients of `a` [H, D, W] / `a_2` [H, D] into the given tensors (e.g. a
    data-parallel gradient bucket's own storage: dist.FlatGradBucket.region) instead of fresh ones; None removes the entry.

    The destination is handed out ONCE per release: the first backward that asks for it claims it, every later one (a second forward
    before any backward — loss(batch1) + loss(batch2), positive / negative passes, a checkpoint re-forward — or gradient accumulation
    over several steps) gets fresh tensors, so that "old + new" never reads the new values on both sides.  The owner releases the claim
    (`release=True`) when it knows that nothing live views the storage any more: SpGAT.fused_head_params does so in a forward that finds
    every parameter's .grad None; `release=False` re-registers the same tensors without touching the claim."""
    if g_a is None:
        _GRAD_DEST.pop(a.data_ptr(), None)
        return
    d = _GRAD_DEST.get(a.data_ptr())
    if d is not None and d.g_a is g_a and d.g_a_2 is g_a_2:
        if release:
            d.claimed = False
        return
    if len(_GRAD_DEST) >= 64:
        _GRAD_DEST.clear()
    d = _GRAD_DEST[a.data_ptr()] = _GradDest(g_a, g_a_2)
    d.claimed = not release


def _weight_grad_tensors(a, H, D, W, f32):
    """Decided at BACKWARD time: the registered destination if nobody has claimed it since its release, else fresh tensors."""
    d = _GRAD_DEST.get(a.data_ptr())
    if d is not None and not d.claimed and d.g_a.shape == (H, D, W) and d.g_a_2.shape == (H, D) and d.g_a.device == a.device \
            and d.g_a.is_contiguous() and d.g_a_2.is_contiguous():
        d.claimed = True
        return d.g_a, d.g_a_2
    return torch.empty(H, D, W, **f32), torch.empty(H, D, **f32)


def set_weight_grad_sync(sync):
    """Install (or, with None, remove) the cross-rank reducer of the heads' weight gradients; returns the previous one."""
    global _WEIGHT_GRAD_SYNC
    prev, _WEIGHT_GRAD_SYNC = _WEIGHT_GRAD_SYNC, sync
    return prev
# The layer's three large products run on split-precision MFMA GEMMs (fp32-accurate) when they are large enough to pay for
# the extra launches (term planes of a, a^T, g_h): measured cross-over on MI355X at cfg-2 widths is 192..256 graphs, i.e.
# ~6 GFLOP per product for bf16 x 3 (~1 GFLOP for f16 x 2, whose operands arrive pre-split).  RECON_GEMM_BX3 (one switch for GAT and GraphConvolution) =
#   auto (default): above the cross-over the f16 x 2 family (csrc/gemm_hx2.hip: 2 half terms per operand under a per-tensor
#                   power-of-two scale, 3 MFMAs per product, operands pre-split by the kernels that produce them) where the
#                   shape allows ((2F+R) % 8 == 0, D % 8 == 0), else bf16 x 3; below it the exact-fp32 MFMA GEMMs
#   2: f16 x 2 always (falls back to bf16 x 3 per shape)   1: bf16 x 3 always (csrc/gemm_bx3.hip)   0: exact fp32 always
_GEMM_BX3 = os.environ.get("RECON_GEMM_BX3", "auto")
_BX3_MIN_FLOP = 6.0e9
_HX2_MIN_FLOP = 1.0e9     # f16 x 2: operands arrive pre-split from their producers, so it pays from smaller products on (cfg 2 at D = 25 -> 32: 2.5 GFLOP, 1.00 -> 0.94 ms per SpGAT step)
SPLIT_BF16X3, SPLIT_F16X2 = 0, 2
_SIDE_STREAMS = {}


def _side_stream(dev):
    s = _SIDE_STREAMS.get(dev)
    if s is None:
        s = _SIDE_STREAMS[dev] = torch.cuda.Stream(device=dev)
    return s


def _require_gpu_f32(*tensors):
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("recon_amd: expected a GPU tensor (this package has no CPU path)")
        if t.dtype != torch.float32:
            raise TypeError("recon_amd: the HIP kernels compute in float32, got %s" % t.dtype)


class ConvKB(nn.Module):
    """Scorer exported by GAT/layers.py:12-48 (not on the hot path; re-exported so that
    `from layers import SpGraphAttentionLayer, ConvKB` keeps working).  Live path: fc1 -> LeakyReLU ->
    fc2; conv_layer / fc_layer / dropout exist only for state_dict compatibility."""

    def __init__(self, input_dim, input_seq_len, in_channels, out_channels, drop_prob, alpha_leaky):
        super().__init__()
        self.conv_layer = nn.Conv2d(in_channels, out_channels, (1, input_seq_len))
        self.dropout = nn.Dropout(drop_prob)
        self.non_linearity = nn.LeakyReLU()
        self.fc_layer = nn.Linear(input_dim * out_channels, 1)
        self.fc1 = nn.Linear(input_dim * 3, input_dim)
        self.nl1 = nn.LeakyReLU()
        self.fc2 = nn.Linear(input_dim, 1)
        nn.init.xavier_uniform_(self.fc_layer.weight, gain=1.414)
        nn.init.xavier_uniform_(self.conv_layer.weight, gain=1.414)

    def forward(self, conv_input):
        return self.fc2(self.nl1(self.fc1(conv_input)))


# ------------------------------------------------------------------------------- G1-G3
class SpecialSpmmFunctionFinal(torch.autograd.Function):
    """out[r] = sum_{e: edge[0,e]==r} edge_w[e]   (GAT/layers.py:51-79).  Backward is the row gather
    grad_out[edge[0]]; no gradient for the indices."""

    @staticmethod
    def forward(ctx, edge, edge_w, N, E, out_features):
        _require_gpu_f32(edge_w)
        L = _lib.lib()
        g = prepare_graph(edge, None, N)
        w = edge_w.contiguous().view(g.E, -1 if g.E else int(out_features))       # no edges: the width comes from the argument
        out = torch.empty(N, w.shape[1], dtype=torch.float32, device=w.device)
        ws = torch.empty(L.recon_spmm_rowsum_workspace_floats(g.E, w.shape[1]), dtype=torch.float32, device=w.device)
        with _lib.on_device(w.device):
            _lib.check(L.recon_spmm_rowsum_fwd(C.byref(g.c), w.data_ptr(), w.shape[1], out.data_ptr(), ws.data_ptr(),
                                               _lib.current_stream()), "recon_spmm_rowsum_fwd")
        ctx.graph = g
        ctx.N, ctx.outfeat, ctx.E = N, w.shape[1], E
        ctx.w_shape = edge_w.shape
        return out

    @staticmethod
    def backward(ctx, grad_output):
        grad_values = None
        if ctx.needs_input_grad[1]:
            g = ctx.graph
            L = _lib.lib()
            go = grad_output.contiguous()
            grad_values = torch.empty(g.E, ctx.outfeat, dtype=torch.float32, device=go.device)
            with _lib.on_device(go.device):
                _lib.check(L.recon_spmm_rowsum_bwd(g.edge[0].data_ptr(), g.E, go.data_ptr(), ctx.outfeat,
                                                   grad_values.data_ptr(), _lib.current_stream()),
                           "recon_spmm_rowsum_bwd")
        return None, (grad_values.view(ctx.w_shape) if grad_values is not None else None), None, None, None


class _GatherRows(torch.autograd.Function):
    """rows = table[index]; the backward is the same segment sum as SpecialSpmmFinal (rows grouped by index,
    fixed order) instead of torch's sort-based index backward."""

    @staticmethod
    def forward(ctx, table, index):
        _require_gpu_f32(table)
        ctx.n_rows = table.shape[0]
        ctx.index = index
        return table.index_select(0, index)

    @staticmethod
    def backward(ctx, grad):
        return _rowsum_by_index(grad.contiguous(), ctx.index, ctx.n_rows), None


class _GatherRowsPair(torch.autograd.Function):
    """table[idx[:, 0]] + table[idx[:, 1]] for idx int64 [E, 2] — the n-hop relation embedding of GAT/models.py:64-65 / :80-81
    (`relation_embed[edge_type_nhop[:, 0]] + relation_embed[edge_type_nhop[:, 1]]`).  One gather-and-add forward; backward ONE fixed-order
    segment sum over the 2 E keys (idx[:, 0] | idx[:, 1]) whose slot k reads gradient row k % E (recon_spmm_rowsum_mod_fwd) — as two
    gather_rows it was two key tensors, two CSR builds, two segment sums and an add per table."""

    @staticmethod
    def forward(ctx, table, idx2):
        _require_gpu_f32(table)
        ctx.n_rows, ctx.idx2 = table.shape[0], idx2
        table = table.contiguous()
        out = torch.empty(idx2.shape[0], table.shape[1], dtype=torch.float32, device=table.device)
        with _lib.on_device(table.device):
            _lib.check(_lib.lib().recon_gather_rows_pair_fwd(table.data_ptr(), idx2.data_ptr(), idx2.shape[0], table.shape[1], out.data_ptr(),
                                                             _lib.current_stream()), "recon_gather_rows_pair_fwd")
        return out

    @staticmethod
    def backward(ctx, grad):
        idx2 = ctx.idx2
        E = idx2.shape[0]
        grad = grad.contiguous()
        g = prepare_graph(_pair_key(idx2), None, ctx.n_rows, rows_only=True)
        out = torch.empty(ctx.n_rows, grad.shape[1], dtype=torch.float32, device=grad.device)
        L = _lib.lib()
        ws = torch.empty(L.recon_spmm_rowsum_workspace_floats(g.E, grad.shape[1]), dtype=torch.float32, device=grad.device)
        with _lib.on_device(grad.device):
            _lib.check(L.recon_spmm_rowsum_mod_fwd(C.byref(g.c), grad.data_ptr(), grad.shape[1], E, out.data_ptr(), ws.data_ptr(), _lib.current_stream()),
                       "recon_spmm_rowsum_mod_fwd")
        return out, None


def _pair_key(idx2):
    """[2, 2 E] segment key (idx[:, 0] | idx[:, 1]) of an [E, 2] index tensor, one per tensor (identity + version): the two tables of a SpGAT
    forward share it, and with it the CSR that prepare_graph caches."""
    k = ("pair", idx2.data_ptr(), idx2._version, tuple(idx2.shape), tuple(idx2.stride()))
    hit = _KEY_CACHE.get(k)
    if hit is None:
        if len(_KEY_CACHE) > 16:
            _KEY_CACHE.clear()
        flat = idx2.t().reshape(1, -1)
        key = flat.expand(2, -1)                                         # (both rows are the one row: a destination-only graph reads row 0 alone — no copy)
        if trusted(idx2):
            trust(key, bound=trust_bounds(idx2)[0])
        hit = _KEY_CACHE[k] = (key, idx2)
    return hit[0]


def gather_rows_pair(table, idx2):
    """table[idx2[:, 0]] + table[idx2[:, 1]] with a deterministic single-pass backward (see _GatherRowsPair)."""
    if idx2.numel() and _VALIDATE_PAIR and not trusted(idx2, table.shape[0]):
        lo, hi = torch.aminmax(idx2)
        if int(lo) < 0 or int(hi) >= table.shape[0]:
            raise IndexError("recon_amd: row index out of range: [%d, %d] into a table of %d rows" % (int(lo), int(hi), table.shape[0]))
    return _GatherRowsPair.apply(table, idx2.long().contiguous())


_VALIDATE_PAIR = True


class _SmallMM(torch.autograd.Function):
    """A @ B for the models' small dense products (relation_embed.mm(W), GAT/models.py:75: 64..237 rows) on a kernel made for them
    (csrc/gemm_f32.hip k_gemm_small): the library GEMM behind torch.mm takes ~40 us for these 10-MFLOP products on MI355X (rocprofv3, full SpGAT step: 4 calls =
    7 % of a 2 ms step), this one a launch."""

    @staticmethod
    def forward(ctx, A, B):
        _require_gpu_f32(A, B)
        A, B = A.contiguous(), B.contiguous()
        ctx.save_for_backward(A, B)
        return _sgemm_small(A, False, B, False)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        g = g.contiguous()
        gA = _sgemm_small(g, False, B, True) if ctx.needs_input_grad[0] else None      # g B^T: B [K,N] read as the [N',K'] form
        gB = _sgemm_small(A, True, g, False) if ctx.needs_input_grad[1] else None      # A^T g: A [M,K] read as the k-major form
        return gA, gB


class _ThinWeightMM(torch.autograd.Function):
    """A @ B for a tall A [rows, K] and a small weight B [K, N] (`entity_embeddings.mm(self.W_entities)`, GAT/models.py:177).  Forward and
    g_A = g B^T on the fp32 matrix-core GEMM of this library (recon_sgemm_ex); the weight gradient A^T g — a K x N output over `rows`
    terms, for which a library GEMM picks a 32 x 32 x 256 tile kernel that takes 98 us at 14 541 x 50 x 200 — on recon_sgemm_small
    with the long dimension cut over workgroups (fixed-order combine)."""

    @staticmethod
    def forward(ctx, A, B):
        _require_gpu_f32(A, B)
        A, B = A.contiguous(), B.contiguous()
        ctx.save_for_backward(A, B)
        return _sgemm_ex(A, False, B, False)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        g = g.contiguous()
        gA = _sgemm_ex(g, False, B, True) if ctx.needs_input_grad[0] else None
        gB = _sgemm_small(A, True, g, False) if ctx.needs_input_grad[1] else None
        return gA, gB


class _GemmMM(torch.autograd.Function):
    """torch.mm semantics for fp32 GPU matrices of any size on this library's fp32 matrix-core GEMM (recon_sgemm_ex: exact fp32
    v_mfma_f32_32x32x2_f32 tiles, split-K with a fixed-order second pass where the output has few tiles), forward and both gradients."""

    @staticmethod
    def forward(ctx, A, B):
        _require_gpu_f32(A, B)
        A, B = A.contiguous(), B.contiguous()
        ctx.save_for_backward(A, B)
        return _sgemm_ex(A, False, B, False)

    @staticmethod
    def backward(ctx, g):
        A, B = ctx.saved_tensors
        g = g.contiguous()
        gA = _sgemm_ex(g, False, B, True) if ctx.needs_input_grad[0] else None        # g B^T
        gB = _sgemm_ex(A, True, g, False) if ctx.needs_input_grad[1] else None        # A^T g
        return gA, gB


def _sgemm_ex(A, a_is_km, B, b_is_nk):
    """op(A) op(B) on recon_sgemm_ex; A, B contiguous 2-d fp32."""
    M, K = (A.shape[1], A.shape[0]) if a_is_km else A.shape
    N = B.shape[0] if b_is_nk else B.shape[1]
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    if M and N:
        if K == 0:
            return out.zero_()
        L = _lib.lib()
        nws = L.recon_sgemm_ex_workspace_floats(M, N, K)
        ws = torch.empty(nws, dtype=torch.float32, device=A.device) if nws else None
        with _on_device(A.device):
            _lib.check(L.recon_sgemm_ex(M, N, K, A.data_ptr(), A.shape[1], 1 if a_is_km else 0, B.data_ptr(), B.shape[1],
                                        1 if b_is_nk else 0, out.data_ptr(), N, _lib.ptr(ws), _lib.current_stream()), "recon_sgemm_ex")
    return out


def _sgemm_small(A, a_is_km, B, b_is_nk):
    """op(A) op(B) on recon_sgemm_small; A, B contiguous 2-d fp32."""
    M, K = (A.shape[1], A.shape[0]) if a_is_km else A.shape
    N = B.shape[0] if b_is_nk else B.shape[1]
    out = torch.empty(M, N, dtype=torch.float32, device=A.device)
    if M and N:
        L = _lib.lib()
        nws = L.recon_sgemm_small_workspace_floats(M, N, K)
        ws = torch.empty(nws, dtype=torch.float32, device=A.device) if nws else None
        with _on_device(A.device):
            _lib.check(L.recon_sgemm_small(M, N, K, A.data_ptr(), A.shape[1], 1 if a_is_km else 0, B.data_ptr(), B.shape[1],
                                           1 if b_is_nk else 0, out.data_ptr(), N, _lib.ptr(ws), _lib.current_stream()), "recon_sgemm_small")
    return out


_SMALL_MM_FLOP = 2.0e8
_THIN_WEIGHT_ELEMS = 1 << 16        # weights up to 64 k elements: their gradient has at most 256 output tiles


def small_mm(A, B):
    """torch.mm semantics for the models' dense products outside the attention layer; fp32 GPU matrices never reach a library GEMM:
    products below ~0.2 GFLOP run on recon_sgemm_small (_SmallMM), tall-times-small-weight ones on _ThinWeightMM, everything else on
    the fp32 matrix-core GEMM (_GemmMM).  Other dtypes / devices: torch.mm."""
    if (A.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32 and A.dim() == 2 and B.dim() == 2
            and 0 < A.shape[1] == B.shape[0] and 2.0 * A.shape[0] * A.shape[1] * B.shape[1] < _SMALL_MM_FLOP):
        return _SmallMM.apply(A, B)
    if (A.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32 and A.dim() == 2 and B.dim() == 2 and A.shape[1] == B.shape[0]
            and 0 < B.shape[0] * B.shape[1] <= _THIN_WEIGHT_ELEMS and A.shape[0] >= 2048):
        return _ThinWeightMM.apply(A, B)
    if A.is_cuda and A.dtype == torch.float32 and B.dtype == torch.float32 and A.dim() == 2 and B.dim() == 2 and A.shape[1] == B.shape[0]:
        return _GemmMM.apply(A, B)
    return torch.mm(A, B)


_KEY_CACHE = {}


def _segment_key(index):
    """[2,E] edge-style key (row 0 = segment id; row 1 is ignored by the row sum) for an index tensor; one key
    tensor per index tensor (identity + version) so that prepare_graph's CSR cache hits across steps."""
    k = (index.data_ptr(), index._version, tuple(index.shape), tuple(index.stride()))
    hit = _KEY_CACHE.get(k)
    if hit is None:
        if len(_KEY_CACHE) > 16:
            _KEY_CACHE.clear()
        key = index.reshape(1, -1).expand(2, -1)                       # (a destination-only graph reads row 0 alone: no copy)
        if trusted(index):
            trust(key, bound=trust_bounds(index)[0])
        hit = _KEY_CACHE[k] = (key, index)      # keeps `index` alive: its data_ptr is the key
    return hit[0]


def gather_rows(table, index):
    """table[index] with a deterministic, sort-free backward (used for `relation_embed[edge_type]`, GAT/models.py:79)."""
    return _GatherRows.apply(table, index)


class SpecialSpmmFinal(nn.Module):
    def forward(self, edge, edge_w, N, E, out_features):        # GAT/layers.py:82-84
        return SpecialSpmmFunctionFinal.apply(edge, edge_w, N, E, out_features)


# ------------------------------------------------------------------------------- G4 (+ H heads)
def _fwd_args(graph, x, ee, a, a2, keep, P, Q, sigma, Z, out, alpha, concat):
    H, D = a2.shape
    return _lib.GatFwdArgs(graph.N, graph.E, x.shape[1], ee.shape[1], D, H, int(bool(concat)), float(alpha),
                           x.data_ptr(), ee.data_ptr(), a.data_ptr(), a2.data_ptr(), _lib.ptr(keep),
                           P.data_ptr(), Q.data_ptr(), _lib.ptr(sigma), _lib.ptr(Z), out.data_ptr(), out.shape[1])


class _GATHeadsFunction(torch.autograd.Function):
    """H attention heads sharing (x, edges, edge_embed): out[:, h*D:(h+1)*D] = head h.
    One C call for the forward (projection GEMMs + fused edge kernel), one for the backward."""

    @staticmethod
    def forward(ctx, x, ee, a, a2, graph, keep, alpha, concat, keep_iid=False):
        _require_gpu_f32(x, ee, a, a2, keep)
        L = _lib.lib()
        x, ee, a, a2 = x.contiguous(), ee.contiguous(), a.contiguous(), a2.contiguous()
        H, D = a2.shape
        N, E = graph.N, graph.E
        if x.shape[0] != N or ee.shape[0] != E or a.shape != (H, D, 2 * x.shape[1] + ee.shape[1]):
            raise ValueError("recon_amd.gat_heads: inconsistent shapes")
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        need_grad = any(ctx.needs_input_grad[:4])
        P = torch.empty(2, H, N, D, **f32)
        Q = torch.empty(H, E, D, **f32)
        out = torch.empty(N, H * D, **f32)
        sigma = torch.empty(H, E, **f32) if need_grad else None
        Z = torch.empty(H, N, **f32) if need_grad else None
        if keep is not None:
            if not need_grad:
                sigma, Z = torch.empty(H, E, **f32), torch.empty(H, N, **f32)
            # original order -> CSR-slot order; independent draws (keep_iid) are as good in any order: taken as they lie
            keep = keep.view(H, E) if keep_iid else keep.view(H, E)[:, graph.eid_long].contiguous()
        args = _fwd_args(graph, x, ee, a, a2, keep, P, Q, sigma, Z, out, alpha, concat)
        with _lib.on_device(dev):
            _lib.check(L.recon_gat_fwd(C.byref(graph.c), C.byref(args), _lib.current_stream()), "recon_gat_fwd")
        if need_grad:
            ctx.save_for_backward(x, ee, a, a2, keep, P, Q, sigma, Z, out)
            ctx.graph, ctx.alpha, ctx.concat = graph, alpha, concat
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, ee, a, a2, keep, P, Q, sigma, Z, out = ctx.saved_tensors
        graph = ctx.graph
        L = _lib.lib()
        H, D = a2.shape
        N, E, F_, R = graph.N, graph.E, x.shape[1], ee.shape[1]
        dev = x.device
        f32 = dict(dtype=torch.float32, device=dev)
        grad_out = grad_out.contiguous()
        nx, ne, na, na2 = ctx.needs_input_grad[:4]
        Gm = torch.empty(H, E, D, **f32)
        gP = torch.empty(2, H, N, D, **f32)
        partial = torch.empty(L.recon_gat_bwd_partial_floats(N, E, F_, R, D, H), **f32)
        g_x = torch.empty(N, F_, **f32) if nx else None
        g_ee = torch.empty(E, R, **f32) if ne else None
        g_a = torch.empty(H, D, 2 * F_ + R, **f32) if na else None
        g_a2 = torch.empty(H, D, **f32) if na2 else None
        args = _lib.GatBwdArgs(_fwd_args(graph, x, ee, a, a2, keep, P, Q, sigma, Z, out, ctx.alpha, ctx.concat),
                               grad_out.data_ptr(), grad_out.shape[1], Gm.data_ptr(), gP.data_ptr(),
                               partial.data_ptr(), _lib.ptr(g_x), _lib.ptr(g_ee), _lib.ptr(g_a), _lib.ptr(g_a2))
        with _lib.on_device(dev):
            _lib.check(L.recon_gat_bwd(C.byref(graph.c), C.byref(args), _lib.current_stream()), "recon_gat_bwd")
        return g_x, g_ee, g_a, g_a2, None, None, None, None, None


def _p(t):
    """Device pointer of a tensor, an int that already is one, or None."""
    return t if (t is None or isinstance(t, int)) else t.data_ptr()


def _atp_args(graph, x, ee, a, a2, keep, u, c_node, c_rel, V, sigma, Z, Zk, out, alpha, concat, a_split=None, aux=None,
              keep_max=1.0, ee_index=None, io_bf16=False):
    """recon_gat_atp_args from tensors or raw device pointers (workspace slices)."""
    H, D = a2.shape
    return _lib.GatAtpArgs(graph.N, graph.E, x.shape[1], ee.shape[1], D, H, int(bool(concat)), float(alpha),
                           x.data_ptr(), ee.data_ptr(), a.data_ptr(), a2.data_ptr(), _p(keep), _p(u),
                           _p(c_node), _p(c_rel), _p(V), _p(sigma), _p(Z), _p(Zk),
                           _p(out), out.shape[1] if out is not None else H * D, _p(a_split), SPLIT_F16X2 if aux is not None else SPLIT_BF16X3,
                           float(keep_max), _p(aux), _p(ee_index), ee.shape[0] if ee_index is not None else 0, int(bool(io_bf16)))


_PAD_MIN_OUT = 1 << 18            # below this many output elements the padding's extra launches cost more than the aligned GEMMs win


def _atp_split_mode(F_, R, D, H, N=None):
    """0: exact-fp32 MFMA GEMMs, 1: bf16 x 3, 2: f16 x 2 (see _GEMM_BX3 above)."""
    if _GEMM_BX3 == "0":
        return 0
    flop = 2.0 * N * (2 * F_ + R) * H * D if N is not None else float("inf")
    hx2 = _GEMM_BX3 != "1" and _lib.lib().recon_gat_atp_f16x2_supported(F_, R, D, H) == 1
    if _GEMM_BX3 not in ("1", "2") and flop < (_HX2_MIN_FLOP if hx2 else _BX3_MIN_FLOP):
        return 0
    return 2 if hx2 else 1


def _atp_split_buffer(F_, R, D, H, dev, N=None):
    """(a_split, aux): workspace of the split-precision GEMMs (term planes of a and a^T) and, for the f16 x 2 family, the
    block of max-magnitude slots + zero page; (None, None) to stay on the fp32-MFMA GEMMs (RECON_GEMM_BX3=0, or a
    product too small to pay for the extra launches)."""
    mode = _atp_split_mode(F_, R, D, H, N)
    if mode == 0:
        return None, None
    a_split = torch.empty(_lib.lib().recon_gat_atp_split_bytes(F_, R, D, H), dtype=torch.uint8, device=dev)
    aux = None
    if mode == 2:
        aux = torch.empty(_lib.lib().recon_hx2_aux_bytes(), dtype=torch.uint8, device=dev)     # allocator blocks are 512-byte aligned
        if aux.data_ptr() % 256:
            aux = None
    return a_split, aux


def _carve(dev, sizes):
    """ONE allocation for a list of byte sizes (a torch.empty costs 3-4 us of host time, a layer call needs a dozen scratch
    arrays, and at cfg 2 the host side of a step is as long as its GPU side): returns the tensor and the 256-byte aligned
    device pointers of the slices (None for size None)."""
    offs, tot = [], 0
    for sz in sizes:
        if sz is None:
            offs.append(None)
        else:
            offs.append(tot)
            tot += (int(sz) + 255) & ~255
    buf = torch.empty(tot + 256, dtype=torch.uint8, device=dev)
    base = (buf.data_ptr() + 255) & ~255
    return buf, [None if o is None else base + o for o in offs]


def _view_f32(buf, ptr, n):
    """float32 [n] tensor over the slice of a _carve() buffer that starts at device pointer `ptr` (slices are 256-byte aligned)."""
    off = int(ptr) - buf.data_ptr()
    return buf[off:off + 4 * n].view(torch.float32)


_SIZE_CACHE = {}
_GEE16 = {}


def _gee_bf16_ok(F_, R, D, H):
    k = (F_, R, D, H)
    v = _GEE16.get(k)
    if v is None:
        v = _GEE16[k] = _lib.lib().recon_gat_atp_bwd_gee_bf16_supported(F_, R, D, H) == 1
    return v


def _lib_sizes(N, E, F_, R, D, H):
    """Workspace sizes of the C ABI for one shape (ctypes calls are not free either: cached per shape)."""
    key = (N, E, F_, R, D, H)
    v = _SIZE_CACHE.get(key)
    if v is None:
        L = _lib.lib()
        if len(_SIZE_CACHE) > 64:
            _SIZE_CACHE.clear()
        v = _SIZE_CACHE[key] = (L.recon_gat_atp_split_bytes(F_, R, D, H), L.recon_hx2_aux_bytes(),
                                4 * L.recon_gat_atp_bwd_partial_floats(N, E, F_, R, D, H),
                                4 * L.recon_gat_atp_bwd_partial2_floats(N, E, F_, R, D, H), L.recon_gat_atp_bwd_split_bytes(N, D, H))
    return v


def _on_device(dev):
    """Context that makes `dev` current — only entered when it is not already (the context manager costs ~10 us)."""
    return torch.cuda.device(dev) if torch.cuda.current_device() != dev.index else _NULL_CTX


class _NullCtx:
    def __enter__(self):
        return None

    def __exit__(self, *exc):
        return False


_NULL_CTX = _NullCtx()


class _GATHeadsATPFunction(torch.autograd.Function):
    """Same contract as _GATHeadsFunction, through the aggregate-then-project kernels (csrc/gat_atp.hip)."""

    @staticmethod
    def forward(ctx, x, ee, a, a2, graph, keep, alpha, concat, keep_max, ee_index=None, keep_iid=False):
        """ee_index (int64 [E], original edge order) makes `ee` a table: edge e uses row ee_index[e] (see gat_heads).
        x and ee may both be bfloat16 (gat_heads checks the shape: recon_gat_atp_bf16_io_supported): the forward kernels read them in
        place, as do the backward's; the result and the arithmetic stay float32 and the two input gradients are rounded to bfloat16 at the end."""
        io16 = x.dtype == torch.bfloat16
        if io16:
            if ee.dtype != torch.bfloat16 or ee_index is not None or not (x.is_cuda and ee.is_cuda):
                raise ValueError("recon_amd.gat_heads: bfloat16 features need bfloat16 edge embeddings on the GPU (no table mode)")
            _require_gpu_f32(a, a2, keep)
        else:
            _require_gpu_f32(x, ee, a, a2, keep)
        L = _lib.lib()
        x, ee, a, a2 = x.contiguous(), ee.contiguous(), a.contiguous(), a2.contiguous()
        if io16:                                                    # bfloat16 rows are read 16 bytes at a time: a view at an odd storage offset is copied
            x = x if x.data_ptr() % 16 == 0 else x.clone()
            ee = ee if ee.data_ptr() % 16 == 0 else ee.clone()
        H, D = a2.shape
        N, E = graph.N, graph.E
        F_, R = x.shape[1], ee.shape[1]
        W = 2 * F_ + R
        if x.shape[0] != N or (ee.shape[0] != E and ee_index is None) or a.shape != (H, D, W):
            raise ValueError("recon_amd.gat_heads: inconsistent shapes")
        idx_slot = None
        if ee_index is not None:
            if ee_index.shape != (E,) or ee_index.dtype != torch.int64:
                raise ValueError("recon_amd.gat_heads: ee_index must be an int64 [E] tensor")
            idx_slot = graph.slot_order_index(ee_index, ee.shape[0])          # int32 [E], the row each CSR slot reads
        dev = x.device
        need_grad = any(ctx.needs_input_grad[:4])
        train = need_grad or keep is not None
        mode = 2 if io16 else _atp_split_mode(F_, R, D, H, N)           # bfloat16 inputs: the f16 x 2 family's kernels only
        split_bytes, aux_bytes = _lib_sizes(N, E, F_, R, D, H)[:2]
        # u [H,W], c_node [N,2H], c_rel [E,H], V [N,H,W], sigma [E,H], Z [N,H], Zk [N,H], a_split, aux: one allocation
        ws, (u, c_node, c_rel, V, sigma, Z, Zk, a_split, aux) = _carve(dev, (
            4 * H * W, 4 * N * 2 * H, 4 * max(E, ee.shape[0] if ee_index is not None else 0) * H, 4 * N * H * W, 4 * E * H if train else None, 4 * N * H if train else None,
            4 * N * H if train else None, split_bytes if mode else None, aux_bytes if mode == 2 else None))
        if keep is not None:
            # [H,E] original order -> [E,H] slot order; independent draws (keep_iid: H E factors nobody has tied to edges yet) are read as [E,H]
            keep = keep.reshape(E, H) if keep_iid else keep.view(H, E)[:, graph.eid_long].t().contiguous()
        if keep is None:
            keep_max = 1.0
        elif keep_max is None:                                  # explicit factors without a bound: one host read (tests)
            keep_max = float(keep.max()) if aux is not None and keep.numel() else 1.0
        args = _atp_args(graph, x, ee, a, a2, keep, u, c_node, c_rel, V, sigma, Z, Zk, None, alpha, concat, a_split, aux, keep_max, idx_slot, io16)
        args.out = V                                                     # (placeholder: the score stage writes no output; the real one follows the resolve below)
        # a graph built a moment ago has not read its hub-table sizes back yet (graph.GraphCSR.resolve: a host synchronisation).  The score
        # stage needs none of them: it goes out first, so that the device has work behind the build while the host waits and launches again
        early = graph.pending and graph.build_stream == _lib.current_stream()
        if early:
            with _on_device(dev):
                _lib.check(L.recon_gat_atp_scores(C.byref(graph.raw_struct()), C.byref(args), _lib.current_stream()), "recon_gat_atp_scores")
        gstruct, _hub_keep = graph.call_struct(F_, R, H)
        # few destination rows have edges (graph.ROWS_COMPACT_MAX; known once the graph is resolved): the layer runs over those rows, writes
        # their outputs as a compact [n_rows, H D] block and the block is spread out over zeros below
        NR = graph.n_rows or N
        out = torch.empty(NR, H * D, dtype=torch.float32, device=dev)
        args.out = out.data_ptr()
        with _on_device(dev):
            if early:
                _lib.check(L.recon_gat_atp_aggregate(C.byref(gstruct), C.byref(args), _lib.current_stream()), "recon_gat_atp_aggregate")
                _lib.check(L.recon_gat_atp_project(C.byref(gstruct), C.byref(args), _lib.current_stream()), "recon_gat_atp_project")
            else:
                _lib.check(L.recon_gat_atp_fwd(C.byref(gstruct), C.byref(args), _lib.current_stream()), "recon_gat_atp_fwd")
        if need_grad:
            ctx.save_for_backward(x, ee, a, a2, keep, out, ws)
            ctx.ptrs = (u, c_node, c_rel, V, sigma, Z, Zk, a_split, aux)
            ctx.graph, ctx.alpha, ctx.concat, ctx.keep_max = graph, alpha, concat, keep_max
            ctx.idx_slot = idx_slot
        if NR != N:
            full = torch.empty(N, H * D, dtype=torch.float32, device=dev)
            with _on_device(dev):
                _lib.check(L.recon_rows_expand(out.data_ptr(), H * D, graph.c.node_row, N, H * D, full.data_ptr(), H * D, _lib.current_stream()), "recon_rows_expand")
            return full
        return out

    @staticmethod
    def backward(ctx, grad_out):
        x, ee, a, a2, keep, out, ws = ctx.saved_tensors
        u, c_node, c_rel, V, sigma, Z, Zk, a_split, aux = ctx.ptrs
        graph = ctx.graph
        L = _lib.lib()
        H, D = a2.shape
        N, E, F_, R = graph.N, graph.E, x.shape[1], ee.shape[1]
        W = 2 * F_ + R
        dev = x.device
        io16 = x.dtype == torch.bfloat16                        # bfloat16 rows are read in place by the backward's kernels too (round 5)
        f32 = dict(dtype=torch.float32, device=dev)
        grad_out = grad_out.contiguous()
        nx, ne, na, na2 = ctx.needs_input_grad[:4]
        want_a = na or na2
        _, _, partial_b, partial2_b, ghs_b = _lib_sizes(N, E, F_, R, D, H)
        use_gh_planes = aux is not None or (a_split is not None and want_a and os.environ.get("RECON_GEMM_BX3_KM", "1") != "0")
        # scratch of the backward in one allocation: g_h [N,HD] (not in f16 x 2 mode: g_h exists as half planes only), g_V
        # [N,H,W], g_sigma [E,H], Gxs [E,F], gxd [N,F], Gs [N,2H], g_u [H,W], q [N,H], split-K partials, skinny partials,
        # g_h term planes
        ws2, (g_h, g_V, g_sigma, Gxs, gxd, Gs, g_u, q, partial, partial2, gh_split) = _carve(dev, (
            4 * N * H * D if ((ctx.concat or graph.n_rows) and aux is None) else None, 4 * N * H * W, 4 * E * H, 4 * E * F_, 4 * N * F_, 4 * N * 2 * H,
            4 * H * W, 4 * N * H, partial_b, partial2_b, ghs_b if use_gh_planes else None))
        g_x = torch.empty(N, F_, **f32) if nx else None
        # bfloat16 edge embeddings take a bfloat16 gradient: where one head group per wave walks all heads the edge pass rounds its sums as it
        # stores them (one E x R tensor written in 2-byte values, not an fp32 one written, re-read and cast: 116 + 58 MB at configs[4])
        gee16 = bool(ne and io16 and ctx.idx_slot is None and _gee_bf16_ok(F_, R, D, H))
        g_ee = (torch.empty(E, R, dtype=torch.bfloat16, device=dev) if gee16 else torch.empty(E, R, **f32)) if ne else None
        g_a, g_a2 = _weight_grad_tensors(a, H, D, W, f32) if want_a else (None, None)
        fwd = _atp_args(graph, x, ee, a, a2, keep, u, c_node, c_rel, V, sigma, Z, Zk, out, ctx.alpha, ctx.concat, a_split, aux,
                        ctx.keep_max, ctx.idx_slot, io16)
        args = _lib.GatAtpBwdArgs(fwd, grad_out.data_ptr(), grad_out.shape[1], g_h, g_V, g_sigma, Gxs, gxd, Gs, g_u, q, partial, partial2,
                                  _lib.ptr(g_x), _lib.ptr(g_ee), _lib.ptr(g_a), _lib.ptr(g_a2), gh_split, 1 if gee16 else 0)
        gstruct, _hub_keep = graph.call_struct(F_, R, H)
        sync = _WEIGHT_GRAD_SYNC if g_a is not None else None
        with _on_device(dev):
            if sync is not None:
                # PREPARE -> WEIGHTS (+ split-K sum: g_a = G) -> [all-reduce G, asynchronous] || INPUTS -> all-reduce g_u -> FINISH on the means
                st = _lib.current_stream()
                gc, ac = C.byref(gstruct), C.byref(args)
                _lib.check(L.recon_gat_atp_bwd_phase(gc, ac, 1 | 4 | 16, st), "recon_gat_atp_bwd_phase")
                handle = sync.all_reduce_mean(g_a, async_op=True)
                _lib.check(L.recon_gat_atp_bwd_phase(gc, ac, 2, st), "recon_gat_atp_bwd_phase")
                g_u_t = _view_f32(ws2, g_u, H * W)                       # g_u lives inside ws2: hand the reducer a tensor view of it
                sync.all_reduce_mean(g_u_t, async_op=False)
                if handle is not None:
                    handle.wait()
                _lib.check(L.recon_gat_atp_bwd_phase(gc, ac, 8 | 16, st), "recon_gat_atp_bwd_phase")
                sync.mark_reduced(g_a, g_a2)
            elif _OVERLAP and g_a is not None:
                # PREPARE -> { INPUTS on this stream , WEIGHTS (MFMA-bound GEMM) on a side stream } -> FINISH
                main = torch.cuda.current_stream()
                side = _side_stream(dev)
                gc, ac = C.byref(gstruct), C.byref(args)
                _lib.check(L.recon_gat_atp_bwd_phase(gc, ac, 1, main.cuda_stream), "recon_gat_atp_bwd_phase")
                side.wait_stream(main)
                _lib.check(L.recon_gat_atp_bwd_phase(gc, ac, 4, side.cuda_stream), "recon_gat_atp_bwd_phase")
                _lib.check(L.recon_gat_atp_bwd_phase(gc, ac, 2, main.cuda_stream), "recon_gat_atp_bwd_phase")
                main.wait_stream(side)
                _lib.check(L.recon_gat_atp_bwd_phase(gc, ac, 8, main.cuda_stream), "recon_gat_atp_bwd_phase")
                for t in (grad_out, ws, ws2, g_a, a2, a):                # used on the side stream: keep the allocator honest
                    if t is not None:
                        t.record_stream(side)
            else:
                _lib.check(L.recon_gat_atp_bwd(C.byref(gstruct), C.byref(args), _lib.current_stream()), "recon_gat_atp_bwd")
        if g_ee is not None and ctx.idx_slot is not None:
            # table mode: g_ee holds one row per CSR slot; the table's gradient is their sum by row index (fixed order: the same
            # segment walk as SpecialSpmmFinal)
            g_ee = _rowsum_by_index(g_ee, graph.slot_index_long(ctx.idx_slot), ee.shape[0])
        if io16:
            g_x = g_x.to(torch.bfloat16) if g_x is not None else None
            g_ee = g_ee.to(torch.bfloat16) if (g_ee is not None and g_ee.dtype != torch.bfloat16) else g_ee
        return g_x, g_ee, (g_a if na else None), (g_a2 if na2 else None), None, None, None, None, None, None, None


def _rowsum_by_index(rows, index, n_rows):
    """out[r] = sum of rows[k] over k with index[k] == r (index int64 [E]), fixed summation order."""
    g = prepare_graph(_segment_key(index), None, n_rows, rows_only=True)
    out = torch.empty(n_rows, rows.shape[1], dtype=torch.float32, device=rows.device)
    L = _lib.lib()
    ws = torch.empty(L.recon_spmm_rowsum_workspace_floats(g.E, rows.shape[1]), dtype=torch.float32, device=rows.device)
    with _lib.on_device(rows.device):
        _lib.check(L.recon_spmm_rowsum_fwd(C.byref(g.c), rows.data_ptr(), rows.shape[1], out.data_ptr(), ws.data_ptr(), _lib.current_stream()),
                   "recon_spmm_rowsum_fwd")
    return out


def _rowsum_keyed(rows, key, n_rows):
    """_rowsum_by_index for a caller that already holds the [2, E] segment key (losses.py: written by its forward kernel)."""
    g = prepare_graph(key, None, n_rows, rows_only=True)
    out = torch.empty(n_rows, rows.shape[1], dtype=torch.float32, device=rows.device)
    L = _lib.lib()
    ws = torch.empty(L.recon_spmm_rowsum_workspace_floats(g.E, rows.shape[1]), dtype=torch.float32, device=rows.device)
    with _lib.on_device(rows.device):
        _lib.check(L.recon_spmm_rowsum_fwd(C.byref(g.c), rows.data_ptr(), rows.shape[1], out.data_ptr(), ws.data_ptr(), _lib.current_stream()),
                   "recon_spmm_rowsum_fwd")
    return out


_NAN_FLAGS = {}


def enable_nan_flag(device):
    """Give the attention kernels of `device` a word to raise when a row sum of attention weights comes out NaN / infinite — the condition
    behind the reference's `assert not torch.isnan(...)` on edge_e, e_rowsum and h_prime (GAT/layers.py:147, :167, :172) — without their
    host round trips.  Read it with nan_raised() whenever convenient (end of an epoch, next to loss.item()).  Returns the flag tensor."""
    device = torch.device(device)
    idx = device.index if device.index is not None else torch.cuda.current_device()
    f = _NAN_FLAGS.get(idx)
    if f is None:
        f = _NAN_FLAGS[idx] = torch.zeros(1, dtype=torch.int32, device=torch.device("cuda", idx))
        _lib.check(_lib.lib().recon_set_nan_flag(idx, f.data_ptr()), "recon_set_nan_flag")
    return f


def nan_raised(device, reset=True):
    """Whether a kernel has raised the device's NaN word since the last reset (one host round trip).  Enables the word on first use."""
    f = enable_nan_flag(device)
    hit = bool(int(f.item()))
    if hit and reset:
        f.zero_()
    return hit


_ONES = {}


def _ones(n, device):
    """A view of n ones (one read-only buffer per device, grown as needed): draw_keep asks for one per head and forward — a fill launch each."""
    buf = _ONES.get(device)
    if buf is None or buf.numel() < n:
        buf = _ONES[device] = torch.ones(max(n, 1 << 16), dtype=torch.float32, device=device)
    return buf[:n]


def gat_path_for(N, E, F_, R, D, H):
    """Which formulation gat_heads uses: 'atp' (aggregate, then project) when instantiated for the shape and
    the graph is not much sparser than its node set (or its rows with edges get compacted), else 'proj' (project, then aggregate).  RECON_GAT_PATH /
    `gat_layers._GAT_PATH` = 'atp' | 'proj' forces one."""
    if _GAT_PATH == "proj":
        return "proj"
    ok = _lib.lib().recon_gat_atp_supported(N, E, F_, R, D, H) == 1        # a function of the widths only: the same on every rank
    if _GAT_PATH == "atp" or _data_parallel():
        # Data parallelism: every rank has its own batch, so a rule that looks at E / N could send one rank down 'atp' (whose backward
        # issues two collectives under an OverlappedWeightGradSync) and another down 'proj' (which issues none): mismatched collectives
        # hang.  With more than one rank the choice depends on the layer widths alone.
        return "atp" if ok else "proj"
    # few edges for the node set: 'proj' projects every node, 'atp' walks every node's (empty) row — unless the graph's rows get compacted
    # (graph.ROWS_COMPACT_MAX: resolved graphs of more than HUB_CHUNK edges), when 'atp' costs what the rows WITH edges cost
    from . import graph as _graph
    compacts = _graph.ROWS_COMPACT_MAX >= 0.5 and _graph.HUB_CHUNK > 0 and E > _graph.HUB_CHUNK
    return "atp" if (ok and (2 * E >= N or compacts)) else "proj"


def _data_parallel():
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1


def gat_heads(x, edge_embed_all, a, a_2, graph, keep=None, alpha=0.2, concat=True, keep_max=None, ee_index=None, keep_iid=False):
    """Fused forward of H `SpGraphAttentionLayer`s that share their inputs (GAT/models.py:71-72).

    x [N,F]; edge_embed_all [E,R] (1-hop rows then n-hop rows, original order); a [H,D,2F+R];
    a_2 [H,D]; graph = prepare_graph(edge, edge_list_nhop, N); keep [H,E] dropout factors in
    original edge order or None; keep_max an upper bound of them (1/(1-p); read back from `keep` when omitted); keep_iid: the H E
    factors are independent draws that nobody has tied to particular edges (draw_keeps_iid) — the kernels take them in the order they lie
    instead of through a gather into CSR-slot order.
    ee_index (int64 [E], original edge order) turns edge_embed_all into a TABLE [T,R]: edge e uses row ee_index[e] —
    `relation_embed[edge_type]` (GAT/models.py:79, :156) read in place by the kernels instead of materialised as E x R by the caller
    (at 272 k edges x 200 columns that tensor is 218 MB, written once and read four times per step); the table's gradient comes back
    summed over the edges of each row.
    Returns [N, H*D] (heads concatenated along dim 1)."""
    H, D = a_2.shape
    if ee_index is not None and (x.dtype != torch.float32 or edge_embed_all.dtype != torch.float32 or
                                 gat_path_for(graph.N, graph.E, x.shape[1], edge_embed_all.shape[1], D, H) != "atp"):
        # only the fp32 ATP kernels read a table through an index: materialise the rows (gather_rows is an fp32 kernel with a fixed-order
        # backward; reduced-precision tables go through index_select, whose autograd is native)
        rows = gather_rows(edge_embed_all, ee_index) if edge_embed_all.dtype == torch.float32 else edge_embed_all.index_select(0, ee_index)
        return gat_heads(x, rows, a, a_2, graph, keep, alpha, concat, keep_max, keep_iid=keep_iid)
    if (x.dtype == torch.bfloat16 and edge_embed_all.dtype == torch.bfloat16 and ee_index is None and x.is_cuda and D % 8 == 0 and
            gat_path_for(graph.N, graph.E, x.shape[1], edge_embed_all.shape[1], D, H) == "atp" and
            _lib.lib().recon_gat_atp_bf16_io_supported(x.shape[1], edge_embed_all.shape[1], D, H) == 1 and _GEMM_BX3 in ("auto", "2")):
        # bfloat16 at the layer's boundary, read IN PLACE by the forward's kernels (csrc/gat_atp.hip, io_bf16): no up-cast copies of x and of the
        # E x R edge embeddings (174 MB of traffic at BASELINE.json configs[4]'s 145 k edges); arithmetic and parameters stay float32
        out = _GATHeadsATPFunction.apply(x, edge_embed_all, a.float(), a_2.float(), graph, keep, alpha, concat, keep_max, None, keep_iid)
        return out.to(x.dtype)
    if x.dtype in (torch.bfloat16, torch.float16):
        # Reduced-precision STORAGE at the layer boundary (BASELINE.json configs[4]: "mixed GAT+Propagation stack, bf16"): features
        # and edge embeddings arrive and leave in x.dtype; scores, softmax, aggregation and projections run the fp32 kernels (the
        # reference's own arithmetic is fp32; its exp(-leakyrelu(.)) has no max subtraction and would overflow half precision).
        out = gat_heads(x.float(), edge_embed_all.float(), a.float(), a_2.float(), graph, keep, alpha, concat, keep_max, keep_iid=keep_iid)
        return out.to(x.dtype)
    if gat_path_for(graph.N, graph.E, x.shape[1], edge_embed_all.shape[1], D, H) == "atp":
        Dp = (D + 7) // 8 * 8
        if Dp != D and (graph.n_rows or graph.N) * H * D >= _PAD_MIN_OUT:      # (rows the products run over: graph.n_rows when the rows with edges are compacted)
            # Heads whose width is not a multiple of 8 (the reference's D = 25 per head) run as Dp-wide heads with zero rows appended to
            # `a` and `a_2`: the extra output columns are act(0) = 0 and are dropped, the extra gradient rows are dropped by autograd.
            # Unpadded, the head offsets h * D are not 16-byte aligned and the three products fall back to scalar-load GEMMs (the
            # weight gradient alone: 188 us at cfg 2 with D = 25).
            N = x.shape[0]
            a_p = torch.nn.functional.pad(a, (0, 0, 0, Dp - D))
            a2_p = torch.nn.functional.pad(a_2, (0, Dp - D))
            out_p = _GATHeadsATPFunction.apply(x, edge_embed_all, a_p, a2_p, graph, keep, alpha, concat, keep_max, ee_index, keep_iid)
            return out_p.view(N, H, Dp)[:, :, :D].reshape(N, H * D)
        return _GATHeadsATPFunction.apply(x, edge_embed_all, a, a_2, graph, keep, alpha, concat, keep_max, ee_index, keep_iid)
    return _GATHeadsFunction.apply(x, edge_embed_all, a, a_2, graph, keep, alpha, concat, keep_iid)


class IndexedRows:
    """`table[index]` that has not been materialised: what the reference writes as `relation_embed[edge_type]` (GAT/models.py:79,
    :156), handed to the attention layer as the pair so that the kernels read the table in place (gat_heads' ee_index)."""

    def __init__(self, table, index):
        self.table, self.index = table, index

    def materialize(self):
        return gather_rows(self.table, self.index)


_INDEX_CACHE = OrderedDict()


def _extended_index(index, n_rows, n_extra):
    """cat(index, n_rows + arange(n_extra)): the n-hop edges' rows are appended to the table.  One tensor per (index tensor, sizes) so
    that the per-graph caches keyed on it (slot order, row-sum CSR) hit across steps."""
    key = (index.data_ptr(), index._version, tuple(index.shape), int(n_rows), int(n_extra))
    hit = _INDEX_CACHE.get(key)
    if hit is None:
        ext = torch.cat((index, torch.arange(n_rows, n_rows + n_extra, dtype=torch.int64, device=index.device)))
        if trusted(index, n_rows):
            trust(ext, bound=int(n_rows) + int(n_extra))
        hit = (ext, index)
        _INDEX_CACHE[key] = hit
        while len(_INDEX_CACHE) > 8:
            _INDEX_CACHE.popitem(last=False)
    else:
        _INDEX_CACHE.move_to_end(key)
    return hit[0]


def seed_extended_index(index, n_rows, n_extra, ext):
    """Hand _extended_index the tensor it would build for (index, n_rows, n_extra) — a producer that writes the list anyway (sampler.prune_batch:
    the prune kernel knows every surviving edge's table row) saves the arange + cat."""
    key = (index.data_ptr(), index._version, tuple(index.shape), int(n_rows), int(n_extra))
    if trusted(index, n_rows):
        trust(ext, bound=int(n_rows) + int(n_extra))
    _INDEX_CACHE[key] = (ext, index)
    while len(_INDEX_CACHE) > 8:
        _INDEX_CACHE.popitem(last=False)


def cat_edge_embed(edge_embed, edge_list_nhop, edge_embed_nhop):
    """GAT/layers.py:126-127.  Returns (rows, ee_index): ee_index is None for a materialised [E,R] tensor; for IndexedRows the rows are
    the table (with the n-hop edges' rows appended) and ee_index [E] says which row each edge uses."""
    if isinstance(edge_embed, IndexedRows):
        table, index = edge_embed.table, edge_embed.index
        if _has_nhop(edge_list_nhop):
            return torch.cat((table, edge_embed_nhop), dim=0), _extended_index(index, table.shape[0], edge_embed_nhop.shape[0])
        return table, index
    if _has_nhop(edge_list_nhop):
        return torch.cat((edge_embed, edge_embed_nhop), dim=0), None
    return edge_embed, None


class SpGraphAttentionLayer(nn.Module):
    """Sparse KB-GAT attention layer, GAT/layers.py:87-181 — same constructor, forward signature,
    attributes and parameters (`a` [D, 2F+R], `a_2` [1, D], xavier_normal gain 1.414)."""

    def __init__(self, num_nodes, in_features, out_features, nrela_dim, dropout, alpha, concat=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.num_nodes = num_nodes
        self.alpha = alpha
        self.concat = concat
        self.nrela_dim = nrela_dim
        self.a = nn.Parameter(torch.zeros(size=(out_features, 2 * in_features + nrela_dim)))
        nn.init.xavier_normal_(self.a.data, gain=1.414)
        self.a_2 = nn.Parameter(torch.zeros(size=(1, out_features)))
        nn.init.xavier_normal_(self.a_2.data, gain=1.414)
        self.dropout = nn.Dropout(dropout)
        self.leakyrelu = nn.LeakyReLU(self.alpha)
        self.special_spmm_final = SpecialSpmmFinal()

    def draw_keep(self, E, device):
        """Dropout factors for the E un-normalised attention weights, drawn exactly as the reference
        draws them (one nn.Dropout call on an E-vector, GAT/layers.py:158); None in eval mode."""
        if self.training and self.dropout.p > 0:
            return self.dropout(_ones(E, device)).view(1, E)
        return None

    def plain_draws(self):
        """Whether draw_keep is the method above with a stock nn.Dropout behind it (not replaced by a test that replays recorded factors, not
        a subclass's): only then may a caller draw the factors of several heads in one call (draw_keeps_iid)."""
        return ("draw_keep" not in self.__dict__ and type(self).draw_keep is SpGraphAttentionLayer.draw_keep and type(self.dropout) is nn.Dropout
                and "forward" not in self.dropout.__dict__)

    def draw_keeps_iid(self, H, E, device):
        """H x E independent dropout factors in ONE nn.Dropout call, for callers whose edge list is not the reference's (the pruned batch graphs
        of models.SpKBGATModified: the factors cannot be the reference's edge for edge anyway).  Same distribution as H calls of draw_keep; the
        kernels read them in the order they lie (gat_heads(keep_iid=True)): no cat of the heads' vectors, no gather into slot order."""
        if self.training and self.dropout.p > 0:
            return self.dropout(_ones(H * E, device)).view(H, E)
        return None

    def keep_bound(self):
        """Upper bound of the dropout factors draw_keep produces: 1 / (1 - p)."""
        p = float(self.dropout.p)
        return 1.0 / (1.0 - p) if p < 1.0 else 1.0

    def forward(self, input, edge, edge_embed, edge_list_nhop, edge_embed_nhop, elu=None):
        """elu (extension, default = self.concat as in GAT/layers.py:174-178): apply the ELU inside the layer.  A caller that writes
        `F.elu(layer(...))` around a concat=False layer (GAT/models.py:86) passes elu=True instead and gets the activation — and its
        gradient — from the projection's epilogue rather than from two more passes over the output."""
        N = input.size()[0]                                  # not self.num_nodes (GAT/layers.py:112)
        graph = prepare_graph(edge, edge_list_nhop, N)
        ee, ee_index = cat_edge_embed(edge_embed, edge_list_nhop, edge_embed_nhop)     # edge_embed: [E,R] as in the reference, or IndexedRows
        iid = getattr(self, "iid_keep_draws", False) and self.plain_draws()     # set by a caller whose edge list is its own (models.SpKBGATModified, pruned graphs)
        keep = self.draw_keeps_iid(1, graph.E, input.device) if iid else self.draw_keep(graph.E, input.device)
        out = gat_heads(input, ee, self.a.unsqueeze(0), self.a_2, graph, keep, self.alpha, self.concat if elu is None else bool(elu),
                        keep_max=self.keep_bound() if keep is not None else None, ee_index=ee_index, keep_iid=iid and keep is not None)
        if _DEBUG_NAN:                                       # the reference's asserts (:147,:167,:172), at their price: a host round trip per call
            assert not nan_raised(input.device) and not torch.isnan(out).any()
        return out

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
