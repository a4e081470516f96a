"""GP-GNN propagation step.  The reference has no module for it: the block is inlined four times in
models/models.py (GPGNN :238-277, RECON_EAC :447-487, RECON_EAC_KGGAT :657-701, RECON :895-968).
This module offers the same arithmetic as functions with the reference's tensor conventions, running
in the gfx950 kernels of csrc/prop.hip:

    build_block_adjacency(T, identity, n)            models/models.py:240-259
    propagate(adj_list, h0, nonlinearity, head_indices, tail_indices)     :260-274
    make_start_embedding / get_head_indices / get_tail_indices            utils/embedding_utils.py:170-202
    make_start_entity_embeddings                     utils/context_utils.py:387-426
    build_adjecent_matrix                            utils/build_adjecent_matrix.py:6-17

No CPU path: the tensor functions require GPU tensors.
"""
import ctypes as C
import os
import itertools

import numpy as np
import torch

from . import _lib


def _req(*ts, dtype=torch.float32):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("recon_amd: expected a GPU tensor (this package has no CPU path)")
        if t.is_floating_point() and t.dtype != dtype:
            raise TypeError("recon_amd: this call runs the %s kernels, got a %s tensor" % (dtype, t.dtype))


def _float_dtype(*ts):
    """float32 or bfloat16: the one floating dtype of the call's tensors (the kernels exist for these two)."""
    dts = {t.dtype for t in ts if t is not None and t.is_floating_point()}
    if len(dts) != 1 or next(iter(dts)) not in (torch.float32, torch.bfloat16):
        raise TypeError("recon_amd: the propagation kernels take float32 or bfloat16 tensors of ONE dtype, got %s" % sorted(map(str, dts)))
    return next(iter(dts))


_ZEROS = {}


def _zeros_page(dev):
    """1 KiB of zero bytes per device: what the K tails of the bf16 GEMMs read (include/recon_hip.h: recon_prop_b16_args.zeros)."""
    z = _ZEROS.get(dev)
    if z is None:
        z = _ZEROS[dev] = torch.zeros(1024, dtype=torch.uint8, device=dev)
    return z


# ------------------------------------------------------------------------------- host-side index builders
def _pairs(n):
    return [(i, j) for i in range(n) for j in range(n) if i != j]     # channel order, conversion_util.py:6-20


def make_start_embedding(n, d):
    """[n(n-1), 2dn, 1] float64 numpy array, as utils/embedding_utils.py:170-182 returns it."""
    v = np.zeros((n * (n - 1), 2 * d * n, 1), dtype=np.float64)
    for c, (i, j) in enumerate(_pairs(n)):
        v[c, 2 * d * i: 2 * d * i + d, 0] = 1.0
        v[c, 2 * d * j + d: 2 * d * (j + 1), 0] = 1.0
    return v


def get_head_indices(n, d, bs=50):
    """utils/embedding_utils.py:184-192 (nested lists of ranges there; an int64 array here)."""
    rows = np.asarray([list(range(2 * d * i, 2 * d * (i + 1))) for i, _ in _pairs(n)], dtype=np.int64)
    return np.tile(rows[None], (bs, 1, 1))


def get_tail_indices(n, d, bs=50):
    """utils/embedding_utils.py:194-202."""
    rows = np.asarray([list(range(2 * d * j, 2 * d * (j + 1))) for _, j in _pairs(n)], dtype=np.int64)
    return np.tile(rows[None], (bs, 1, 1))


def build_adjecent_matrix(n, size=72):
    """utils/build_adjecent_matrix.py:6-17 -> torch.FloatTensor [72,72] (row-normalised line graph)."""
    A = np.zeros((size, size), dtype=np.float32)
    V = list(itertools.permutations(range(n), 2))
    for i, x in enumerate(V):
        for j, y in enumerate(V):
            A[i, j] = 1.0 if (x[0] == y[1] or x[1] == y[0] or i == j) else 0.0
        s = A[i].sum()
        if s != 0:
            A[i] /= s
    return torch.from_numpy(A)


# ------------------------------------------------------------------------------- P1
class _BlockAdjacency(torch.autograd.Function):
    @staticmethod
    def forward(ctx, T, identity, n):
        _req(T, identity)
        T, identity = T.contiguous(), identity.contiguous()
        B = T.shape[0]
        dd = identity.shape[0]
        if T.numel() != B * n * (n - 1) * dd * dd:
            raise ValueError("T must hold B x n(n-1) transition matrices of %dx%d" % (dd, dd))
        A = torch.empty(B, n * dd, n * dd, dtype=torch.float32, device=T.device)
        with _lib.on_device(T.device):
            _lib.check(_lib.lib().recon_block_adjacency_fwd(T.data_ptr(), identity.data_ptr(), B, n, dd, A.data_ptr(),
                                                            _lib.current_stream()), "recon_block_adjacency_fwd")
        ctx.dims = (B, n, dd, tuple(T.shape))
        return A

    @staticmethod
    def backward(ctx, gA):
        B, n, dd, tshape = ctx.dims
        gA = gA.contiguous()
        gT = torch.empty(tshape, dtype=torch.float32, device=gA.device) if ctx.needs_input_grad[0] else None
        gI = torch.empty(dd, dd, dtype=torch.float32, device=gA.device) if ctx.needs_input_grad[1] else None
        L = _lib.lib()
        ws = (torch.empty(L.recon_block_adjacency_bwd_workspace_floats(B, n, dd), dtype=torch.float32, device=gA.device)
              if gI is not None else None)
        with _lib.on_device(gA.device):
            _lib.check(L.recon_block_adjacency_bwd(gA.data_ptr(), B, n, dd, _lib.ptr(gT), _lib.ptr(gI), _lib.ptr(ws),
                                                   _lib.current_stream()), "recon_block_adjacency_bwd")
        return gT, gI, None


class _BlockAdjacencyB16(torch.autograd.Function):
    """P1 on bfloat16 tensors (csrc/prop_b16.hip)."""

    @staticmethod
    def forward(ctx, T, identity, n):
        _req(T, identity, dtype=torch.bfloat16)
        T, identity = T.contiguous(), identity.contiguous()
        B = T.shape[0]
        dd = identity.shape[0]
        if T.numel() != B * n * (n - 1) * dd * dd:
            raise ValueError("T must hold B x n(n-1) transition matrices of %dx%d" % (dd, dd))
        A = torch.empty(B, n * dd, n * dd, dtype=torch.bfloat16, device=T.device)
        with _lib.on_device(T.device):
            for b0 in range(0, B, _MAX_BATCH):
                nb = min(B, b0 + _MAX_BATCH) - b0
                _lib.check(_lib.lib().recon_block_adjacency_b16_fwd(T[b0:].data_ptr(), identity.data_ptr(), nb, n, dd, A[b0:].data_ptr(),
                                                                    _lib.current_stream()), "recon_block_adjacency_b16_fwd")
        ctx.dims = (B, n, dd, tuple(T.shape))
        return A

    @staticmethod
    def backward(ctx, gA):
        B, n, dd, tshape = ctx.dims
        gA = gA.contiguous()
        if B > _MAX_BATCH:
            raise NotImplementedError("bfloat16 block adjacency backward: more than %d graphs per call" % _MAX_BATCH)
        gT = torch.empty(tshape, dtype=torch.bfloat16, device=gA.device) if ctx.needs_input_grad[0] else None
        gI = torch.empty(dd, dd, dtype=torch.bfloat16, device=gA.device) if ctx.needs_input_grad[1] else None
        L = _lib.lib()
        ws = (torch.empty(L.recon_block_adjacency_b16_bwd_workspace_floats(dd), dtype=torch.float32, device=gA.device)
              if gI is not None else None)
        with _lib.on_device(gA.device):
            _lib.check(L.recon_block_adjacency_b16_bwd(gA.data_ptr(), B, n, dd, _lib.ptr(gT), _lib.ptr(gI), _lib.ptr(ws),
                                                       _lib.current_stream()), "recon_block_adjacency_b16_bwd")
        return gT, gI, None


def build_block_adjacency(T, identity, n):
    """models/models.py:240-259: T [B, n-1, n, (2d)^2] or [B, n(n-1), (2d)^2] (AFTER the non-linearity),
    identity [2d,2d] -> A [B, S, S], A[b, i*2d+r, j*2d+c] = T[b, e(i,j)].view(2d,2d)[r,c], identity on the
    diagonal blocks.  float32 or bfloat16 (both tensors alike)."""
    if _float_dtype(T, identity) == torch.bfloat16:
        return _BlockAdjacencyB16.apply(T, identity, n)
    return _BlockAdjacency.apply(T, identity, n)


# ------------------------------------------------------------------------------- P2
_MAX_BATCH = 65535          # graphs per launch (recon_propagate_* / recon_gcn_*: RECON_ERR_UNSUPPORTED above)


def _ptr_array(tensors):
    arr = (C.c_void_p * len(tensors))()
    for i, t in enumerate(tensors):
        arr[i] = None if t is None else t.data_ptr()
    return arr


def _split_workspace(args, dev):
    """Workspace of the wide-state two-term forward (include/recon_hip.h: recon_prop_args.split_ws); None where that form does not
    exist.  Sets the two fields of `args`; the caller keeps the returned tensor alive across the launch."""
    nbytes = _lib.lib().recon_propagate_ws_bytes(C.byref(args))
    if not nbytes:
        return None
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    args.split_ws, args.split_ws_bytes = ws.data_ptr(), nbytes
    return ws


class _Propagate(torch.autograd.Function):
    @staticmethod
    def forward(ctx, h0, act, head, tail, *adjs):
        _req(h0, head, tail, *adjs)
        L = len(adjs)
        adj_shapes = [tuple(a.shape) for a in adjs]
        adjs = [a.contiguous().view(a.shape[0], a.shape[-2], a.shape[-1]) for a in adjs]   # accepts [B,1,S,S]
        B, S = adjs[0].shape[0], adjs[0].shape[-1]
        h0c = h0.contiguous()
        if h0c.dim() == 4:            # [B, C, S, 1]   (RECON*: models/models.py:470)
            Cn = h0c.shape[1]
            h0_bs = Cn * S
        else:                         # [C, S, 1]      (GPGNN: models/models.py:260)
            Cn = h0c.shape[0]
            h0_bs = 0
        head, tail = head.contiguous(), tail.contiguous()
        dd = head.shape[-1]
        idx_bs = Cn * dd if head.dim() == 3 and head.shape[0] > 1 else 0
        if head.dim() == 3 and head.shape[0] < B and head.shape[0] > 1:
            raise ValueError("head/tail indices hold %d batch rows but the batch has %d" % (head.shape[0], B))
        dev = h0.device
        out = torch.empty(B, Cn, L * dd, dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad)
        hs = torch.empty(L, B, Cn, S, dtype=torch.float32, device=dev) if need else None
        parr = _ptr_array(adjs)
        args = _lib.PropArgs(B, Cn, S, L, dd, _lib.ACT[act], parr, h0c.data_ptr(), h0_bs, head.data_ptr(),
                             tail.data_ptr(), idx_bs, out.data_ptr(), _lib.ptr(hs), None, None, None, None, 0)
        ws = _split_workspace(args, dev)       # wide states (160 < S <= 512): A_l pre-split into half terms, slice by slice
        stats = None
        if need and _lib.lib().recon_propagate_form(C.byref(args)) & 1:
            # two-term f16 kernels: the forward records per graph the max magnitudes of the states and adjacencies, the backward takes its
            # per-tensor scales from them (include/recon_hip.h: recon_prop_args.stats)
            stats = torch.empty(B, 2 * L + 1, dtype=torch.float32, device=dev)
            args.stats = C.cast(stats.data_ptr(), _lib.c_f32p)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_propagate_fwd(C.byref(args), _lib.current_stream()), "recon_propagate_fwd")
        if need:
            ctx.save_for_backward(h0c, head, tail, hs, *adjs)
            ctx.stats = stats
            ctx.meta = (B, Cn, S, L, dd, act, h0_bs, idx_bs, tuple(h0.shape), adj_shapes)
        return out

    @staticmethod
    def backward(ctx, gout):
        h0c, head, tail, hs, *adjs = ctx.saved_tensors
        B, Cn, S, L, dd, act, h0_bs, idx_bs, h0_shape, adj_shapes = ctx.meta
        dev = gout.device
        gout = gout.contiguous()
        if gout.data_ptr() % 16:
            gout = gout.clone()                                         # (a view at an odd offset: the wide-state kernels read 16-byte pieces)
        g_adjs = [torch.empty(B, S, S, dtype=torch.float32, device=dev) if ctx.needs_input_grad[4 + l] else None
                  for l in range(L)]
        g_h = torch.empty(B, Cn, S, dtype=torch.float32, device=dev)
        parr, garr = _ptr_array(adjs), _ptr_array(g_adjs)
        fwd = _lib.PropArgs(B, Cn, S, L, dd, _lib.ACT[act], parr, h0c.data_ptr(), h0_bs, head.data_ptr(), tail.data_ptr(),
                            idx_bs, None, hs.data_ptr(), None, None, _lib.ptr(ctx.stats), None, 0)
        fwd.out = gout.data_ptr()      # unused by the backward; must be non-null for the argument check
        # wide states (S > 160): GP-GNN's block-structured gather indices let the chain of products run on the forward's two-term f16 kernel
        # (all hops in one launch); otherwise both products of a hop are batched fp32 GEMMs
        blk = chain = ws = wide = None
        if S > 160 and dd == 16 and not idx_bs and h0c.data_ptr() % 16 == 0 and all(a.data_ptr() % 16 == 0 for a in adjs):
            blk = _index_blocks(head, tail, dd, S, align=16)
            if blk is not None:
                ws = _split_workspace(fwd, dev)
                nch = _lib.lib().recon_propagate_bwd_chain_ws_floats(C.byref(fwd)) if ws is not None else 0
                chain = torch.empty(nch, dtype=torch.float32, device=dev) if nch else None
        if chain is None:
            nws = _lib.lib().recon_propagate_bwd_ws_floats(C.byref(fwd))
            wide = torch.empty(nws, dtype=torch.float32, device=dev) if nws else None
        args = _lib.PropBwdArgs(fwd, gout.data_ptr(), garr, g_h.data_ptr(), None, None, None, _lib.ptr(wide),
                                blk[0].data_ptr() if chain is not None else None, blk[1].data_ptr() if chain is not None else None, _lib.ptr(chain))
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_propagate_bwd(C.byref(args), _lib.current_stream()), "recon_propagate_bwd")
        g_h0 = None
        if ctx.needs_input_grad[0]:
            g_h0 = (g_h if h0_bs else g_h.sum(0)).view(h0_shape)
        return (g_h0, None, None, None) + tuple(g.view(sh) if g is not None else None for g, sh in zip(g_adjs, adj_shapes))


def _b16_args(B, Cn, S, L, dd, act, adjs, h0c, h0_bs, head, tail, idx_bs, out, hs, trans=None, identity=None):
    return _lib.PropB16Args(B, Cn, S, L, dd, _lib.ACT[act], _ptr_array(adjs) if adjs is not None else None, h0c.data_ptr(), h0_bs,
                            head.data_ptr(), tail.data_ptr(), idx_bs, _lib.ptr(out), _lib.ptr(hs),
                            _ptr_array(trans) if trans is not None else None, _lib.ptr(identity), _zeros_page(h0c.device).data_ptr())


_BLK_IDX = {}


def _index_blocks(head, tail, dd, S, align=8):
    """(head_blk, tail_blk) int32 [C] device tensors when the gather indices are blocks of dd consecutive columns — head[c, x] = head[c, 0] + x,
    likewise tail, every block inside [0, S], head and tail blocks of a channel distinct — shared by the batch; else None.  That is what
    utils/embedding_utils.py:184-202 builds; the bfloat16 backward then forms Y_l in its GEMM epilogues.  One host read per index tensor
    pair (cached on identity + version)."""
    key = (head.data_ptr(), head._version, tail.data_ptr(), tail._version, tuple(head.shape), int(dd), int(S), int(align))
    hit = _BLK_IDX.get(key)
    if hit is None:
        res = None
        if dd % 8 == 0 and (head.dim() == 2 or head.shape[0] == 1):
            h2, t2 = head.reshape(-1, dd), tail.reshape(-1, dd)
            ar = torch.arange(dd, device=head.device)
            ok = ((h2 == h2[:, :1] + ar).all() & (t2 == t2[:, :1] + ar).all() & (h2[:, 0] % align == 0).all() & (t2[:, 0] % align == 0).all()
                  & (h2[:, 0] >= 0).all() & (t2[:, 0] >= 0).all() & (h2[:, 0] + dd <= S).all() & (t2[:, 0] + dd <= S).all()
                  & ((h2[:, 0] - t2[:, 0]).abs() >= dd).all())
            if bool(ok):
                res = (h2[:, 0].to(torch.int32).contiguous(), t2[:, 0].to(torch.int32).contiguous())
        if len(_BLK_IDX) >= 16:
            _BLK_IDX.clear()
        hit = _BLK_IDX[key] = (res, head, tail)                         # pins the tensors while cached
    return hit[0]


class _PropagateB16(torch.autograd.Function):
    """models/models.py:260-274 on bfloat16 tensors (csrc/prop_b16.hip): bf16 storage, fp32 accumulation, every state rounded to bf16 once
    per hop, gradients in bf16."""

    @staticmethod
    def forward(ctx, h0, act, head, tail, want_states, *adjs):
        _req(h0, head, tail, *adjs, dtype=torch.bfloat16)
        L = len(adjs)
        adj_shapes = [tuple(a.shape) for a in adjs]
        adjs = [a.contiguous().view(a.shape[0], a.shape[-2], a.shape[-1]) for a in adjs]
        B, S = adjs[0].shape[0], adjs[0].shape[-1]
        h0c = h0.contiguous()
        if h0c.dim() == 4:
            Cn, h0_bs = h0c.shape[1], h0c.shape[1] * S
        else:
            Cn, h0_bs = h0c.shape[0], 0
        head, tail = head.contiguous(), tail.contiguous()
        dd = head.shape[-1]
        idx_bs = Cn * dd if head.dim() == 3 and head.shape[0] > 1 else 0
        if head.dim() == 3 and head.shape[0] < B and head.shape[0] > 1:
            raise ValueError("head/tail indices hold %d batch rows but the batch has %d" % (head.shape[0], B))
        dev = h0.device
        out = torch.empty(B, Cn, L * dd, dtype=torch.bfloat16, device=dev)
        need = any(ctx.needs_input_grad)
        args = _b16_args(B, Cn, S, L, dd, act, adjs, h0c, h0_bs, head, tail, idx_bs, out, None)
        form = _lib.lib().recon_propagate_b16_form(C.byref(args))
        hs = torch.empty(L, B, Cn, S, dtype=torch.bfloat16, device=dev) if (need or form == 2 or want_states) else None
        args.h_saved = _lib.ptr(hs)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_propagate_b16_fwd(C.byref(args), _lib.current_stream()), "recon_propagate_b16_fwd")
        if need:
            ctx.save_for_backward(h0c, head, tail, hs, *adjs)
            ctx.meta = (B, Cn, S, L, dd, act, h0_bs, idx_bs, tuple(h0.shape), adj_shapes)
        if want_states:                                                   # the states H^1 .. H^L [L, B, C, S] as the forward stored them (no gradient)
            ctx.mark_non_differentiable(hs)
            return out, hs
        return out

    @staticmethod
    def backward(ctx, gout, *_unused):
        h0c, head, tail, hs, *adjs = ctx.saved_tensors
        B, Cn, S, L, dd, act, h0_bs, idx_bs, h0_shape, adj_shapes = ctx.meta
        dev = gout.device
        gout = gout.contiguous()
        if gout.data_ptr() % 16:                 # a contiguous view at an odd storage offset: the kernels read grad_out in 16-byte pieces
            gout = gout.clone()
        g_adjs = [torch.empty(B, S, S, dtype=torch.bfloat16, device=dev) if ctx.needs_input_grad[5 + l] else None for l in range(L)]
        g_h = torch.empty(B, Cn, S, dtype=torch.bfloat16, device=dev)
        ws = torch.empty(B, Cn, S, dtype=torch.bfloat16, device=dev)
        fwd = _b16_args(B, Cn, S, L, dd, act, adjs, h0c, h0_bs, head, tail, idx_bs, gout, hs)     # `out` is unused by the backward
        blk_idx = _index_blocks(head, tail, dd, S) if idx_bs == 0 else None
        args = _lib.PropB16BwdArgs(fwd, gout.data_ptr(), _ptr_array(g_adjs), g_h.data_ptr(), ws.data_ptr(),
                                   _lib.ptr(blk_idx[0]) if blk_idx else None, _lib.ptr(blk_idx[1]) if blk_idx else None, None, None, None, None)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_propagate_b16_bwd(C.byref(args), _lib.current_stream()), "recon_propagate_b16_bwd")
        g_h0 = None
        if ctx.needs_input_grad[0]:
            g_h0 = (g_h if h0_bs else g_h.sum(0, dtype=torch.float32).to(torch.bfloat16)).view(h0_shape)
        return (g_h0, None, None, None, None) + tuple(g.view(sh) if g is not None else None for g, sh in zip(g_adjs, adj_shapes))


def _propagate_b16(adj_list, h0, nonlinearity, head_indices, tail_indices, return_states=False):
    B, S = adj_list[0].shape[0], adj_list[0].shape[-1]
    Cn = h0.shape[1] if h0.dim() == 4 else h0.shape[0]
    probe = _lib.PropB16Args(B, Cn, S, len(adj_list), head_indices.shape[-1], 1, None, None, 0, None, None, 0, None, None, None, None, None)
    aligned = all(a.data_ptr() % 16 == 0 for a in adj_list) and h0.data_ptr() % 16 == 0
    if not aligned or _lib.lib().recon_propagate_b16_form(C.byref(probe)) == 0:
        # shapes the bf16 kernels do not take (S % 8 != 0): bf16 storage around the float32 kernels
        out = _Propagate.apply(h0.float(), nonlinearity, head_indices, tail_indices, *[a.float() for a in adj_list])
        return out.to(torch.bfloat16)
    return _PropagateB16.apply(h0, nonlinearity, head_indices, tail_indices, bool(return_states), *adj_list)


def propagate(adj_list, h0, nonlinearity, head_indices, tail_indices, return_states=False):
    """models/models.py:260-274.  adj_list: L tensors [B,S,S] (or [B,1,S,S] as the reference views them);
    h0 [C,S,1] shared (GPGNN) or [B,C,S,1] per batch (RECON*); nonlinearity 'relu' | 'tanh' | 'linear'
    (model_params.json "non-linear1"); head/tail_indices int64 [C,2d] or [bs,C,2d] as the reference
    stores them.  Returns cat(relation_1..L, -1): [B, C, 2d*L].
    return_states (bfloat16 kernels, one launch's worth of graphs): also the states H^1 .. H^L [L, B, C, S] as the forward rounded and
    stored them — what its backward takes the activation's derivative from (the parity tests compare gradients on exactly these)."""
    if nonlinearity not in _lib.ACT:
        raise NotImplementedError(nonlinearity)
    B = adj_list[0].shape[0] if adj_list else 0
    if return_states:
        if _float_dtype(h0, *adj_list) != torch.bfloat16 or B > _MAX_BATCH:
            raise NotImplementedError("return_states: bfloat16 tensors, at most %d graphs" % _MAX_BATCH)
        res = _propagate_b16(adj_list, h0, nonlinearity, head_indices, tail_indices, return_states=True)
        return res if isinstance(res, tuple) else (res, None)          # None: the shape ran on the float32 kernels behind casts (no bf16 states)
    one = _propagate_b16 if _float_dtype(h0, *adj_list) == torch.bfloat16 else (lambda adjs, h, act, hi, ti: _Propagate.apply(h, act, hi, ti, *adjs))
    if B > _MAX_BATCH:                    # the kernels index graphs with a 16-bit grid dimension; graphs are independent: run slices
        outs = []
        for b0 in range(0, B, _MAX_BATCH):
            sl = slice(b0, min(B, b0 + _MAX_BATCH))
            hs = h0[sl] if h0.dim() == 4 else h0
            hi = head_indices[sl] if head_indices.dim() == 3 and head_indices.shape[0] == B else head_indices
            ti = tail_indices[sl] if tail_indices.dim() == 3 and tail_indices.shape[0] == B else tail_indices
            outs.append(one([a[sl] for a in adj_list], hs, nonlinearity, hi, ti))
        return torch.cat(outs, dim=0)
    return one(adj_list, h0, nonlinearity, head_indices, tail_indices)


class _PropagateBlocks(torch.autograd.Function):
    """Block adjacency + propagation in one pass (models/models.py:240-274 as the reference inlines it): the kernels read
    A_l[b, i*2d+r, j*2d+c] = T_l[b, e(i,j), r, c] (identity on the diagonal) in place, so A_l is never written or read, and the
    backward writes d loss / d T_l in T's own layout and sums the diagonal blocks into d loss / d identity."""

    @staticmethod
    def forward(ctx, h0, identity, act, head, tail, n, *Ts):
        _req(h0, identity, head, tail, *Ts)
        L = len(Ts)
        dd = identity.shape[0]
        B = Ts[0].shape[0]
        S, Cn = n * dd, n * (n - 1)
        t_shapes = [tuple(t.shape) for t in Ts]
        Ts = [t.contiguous().view(B, Cn, dd * dd) for t in Ts]
        identity = identity.contiguous()
        h0c = h0.contiguous()
        h0_bs = Cn * S if h0c.dim() == 4 else 0
        head, tail = head.contiguous(), tail.contiguous()
        idx_bs = Cn * dd if head.dim() == 3 and head.shape[0] > 1 else 0
        dev = h0.device
        out = torch.empty(B, Cn, L * dd, dtype=torch.float32, device=dev)
        need = any(ctx.needs_input_grad)
        hs = torch.empty(L, B, Cn, S, dtype=torch.float32, device=dev) if need else None
        stats = torch.empty(B, 2 * L + 1, dtype=torch.float32, device=dev) if (need and S <= 160) else None
        tarr = _ptr_array(Ts)
        args = _lib.PropArgs(B, Cn, S, L, dd, _lib.ACT[act], None, h0c.data_ptr(), h0_bs, head.data_ptr(), tail.data_ptr(), idx_bs,
                             out.data_ptr(), _lib.ptr(hs), tarr, identity.data_ptr(), _lib.ptr(stats), None, 0)
        ws = _split_workspace(args, dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_propagate_fwd(C.byref(args), _lib.current_stream()), "recon_propagate_fwd (block mode)")
        if need:
            ctx.save_for_backward(h0c, identity, head, tail, hs, stats, *Ts)
            ctx.meta = (B, Cn, S, L, dd, act, h0_bs, idx_bs, tuple(h0.shape), t_shapes)
        return out

    @staticmethod
    def backward(ctx, gout):
        h0c, identity, head, tail, hs, stats, *Ts = ctx.saved_tensors
        B, Cn, S, L, dd, act, h0_bs, idx_bs, h0_shape, t_shapes = ctx.meta
        dev = gout.device
        gout = gout.contiguous()
        if gout.data_ptr() % 16:
            gout = gout.clone()
        g_Ts = [torch.empty(B, Cn, dd * dd, dtype=torch.float32, device=dev) if ctx.needs_input_grad[6 + l] else None for l in range(L)]
        g_I = torch.empty(dd, dd, dtype=torch.float32, device=dev) if ctx.needs_input_grad[1] else None
        g_h = torch.empty(B, Cn, S, dtype=torch.float32, device=dev)
        Lb = _lib.lib()
        tarr, garr = _ptr_array(Ts), _ptr_array(g_Ts)
        fwd = _lib.PropArgs(B, Cn, S, L, dd, _lib.ACT[act], None, h0c.data_ptr(), h0_bs, head.data_ptr(), tail.data_ptr(), idx_bs,
                            gout.data_ptr(), hs.data_ptr(), tarr, identity.data_ptr(), _lib.ptr(stats), None, 0)
        if S > 160:
            # wide states: chain + d T products on the two-term f16 kernels of csrc/prop_hl.hip (block-structured indices: propagate_blocks checked)
            blk = _index_blocks(head, tail, dd, S, align=16)
            sws = _split_workspace(fwd, dev)
            chain = torch.empty(Lb.recon_propagate_bwd_chain_ws_floats(C.byref(fwd)), dtype=torch.float32, device=dev)
            args = _lib.PropBwdArgs(fwd, gout.data_ptr(), None, g_h.data_ptr(), garr, _lib.ptr(g_I), None, None,
                                    blk[0].data_ptr(), blk[1].data_ptr(), chain.data_ptr())
        else:
            ws = torch.empty(Lb.recon_propagate_identity_ws_floats(dd), dtype=torch.float32, device=dev) if g_I is not None else None
            args = _lib.PropBwdArgs(fwd, gout.data_ptr(), None, g_h.data_ptr(), garr, _lib.ptr(g_I), _lib.ptr(ws), None)
        with _lib.on_device(dev):
            _lib.check(Lb.recon_propagate_bwd(C.byref(args), _lib.current_stream()), "recon_propagate_bwd (block mode)")
        g_h0 = None
        if ctx.needs_input_grad[0]:
            g_h0 = (g_h if h0_bs else g_h.sum(0)).view(h0_shape)
        return (g_h0, g_I, None, None, None, None) + tuple(g.view(sh) if g is not None else None for g, sh in zip(g_Ts, t_shapes))


def _blocks_wide_trainable(B, n, dd, h0, L, head, tail):
    """10 < n <= 32 with gradients in float32: the forward reads the transition tensors in place (csrc/prop_hl.hip) and the backward's chain /
    d T products run on its mirror images — for GP-GNN's block-structured gather indices shared by the batch."""
    if head.dim() == 3 and head.shape[0] > 1:
        return False
    Cn, S = n * (n - 1), n * dd
    if _index_blocks(head.contiguous(), tail.contiguous(), dd, S, align=16) is None:
        return False
    probe = _lib.PropArgs(B, Cn, S, max(1, int(L)), dd, 1, None, h0.data_ptr(), Cn * S if h0.dim() == 4 else 0, None, None, 0, None, None, (C.c_void_p * 1)(),
                          16, None, None, 0)
    probe.split_ws_bytes = _lib.lib().recon_propagate_ws_bytes(C.byref(probe))
    probe.split_ws = 1
    return probe.split_ws_bytes > 0 and _lib.lib().recon_propagate_bwd_chain_ws_floats(C.byref(probe)) > 0


def blocks_mode_available(B, n, dd, h0, need_grad=True, L=1, T_list=None):
    """Whether propagate_blocks() runs fused (two-term f16 kernels for the forward and, when gradients are wanted, for the backward:
    2d = 16, n <= 9; forward only — inference — for 10 <= n <= 32; B within one launch).  `L` (hops) and `T_list` (the transition tensors:
    16-byte alignment) make the probe exact: the wide form's LDS budget grows with the hop count."""
    if dd != 16 or n < 2 or n > 32 or B > _MAX_BATCH or B == 0 or os.environ.get("RECON_PROP_BLOCKS", "1") == "0":
        return False
    if T_list is not None and any(t.data_ptr() % 16 for t in T_list):
        return False
    Cn, S = n * (n - 1), n * dd
    probe = _lib.PropArgs(B, Cn, S, max(1, int(L)), dd, 1, None, h0.data_ptr(), Cn * S if h0.dim() == 4 else 0, None, None, 0, None, None, None, None, None, None, 0)
    if n > 10:          # wide states: the forward-only form of csrc/prop_hl.hip reads the transition tensors in place too
        return (not need_grad) and h0.data_ptr() % 16 == 0 and _lib.lib().recon_propagate_ws_bytes(C.byref(probe)) > 0
    form = _lib.lib().recon_propagate_form(C.byref(probe))
    return form == 3 or (form == 1 and not need_grad)


def _blocks_b16_available(B, n, dd, h0, T_list, identity):
    if dd != 16 or n < 2 or n > 32 or B > _MAX_BATCH or B == 0 or os.environ.get("RECON_PROP_BLOCKS", "1") == "0":
        return False
    if h0.data_ptr() % 16 or identity.data_ptr() % 16 or any(T.data_ptr() % 16 or not T.is_contiguous() for T in T_list):
        return False
    probe = _lib.PropB16Args(B, n * (n - 1), n * dd, len(T_list), dd, 1, None, None, 0, None, None, 0, None, None, None, None, None)
    return _lib.lib().recon_propagate_b16_form(C.byref(probe)) in (1, 3)     # the kernels with the state in LDS read T in place


def _propagate_blocks_b16(T_list, identity, n, h0, act, head, tail):
    """Inference: block adjacency + propagation in one launch on bfloat16 tensors; A_l is never written."""
    _req(h0, identity, head, tail, *T_list, dtype=torch.bfloat16)
    L, dd, B = len(T_list), identity.shape[0], T_list[0].shape[0]
    S, Cn = n * dd, n * (n - 1)
    Ts = [t.contiguous().view(B, Cn, dd * dd) for t in T_list]
    identity, h0c = identity.contiguous(), h0.contiguous()
    h0_bs = Cn * S if h0c.dim() == 4 else 0
    head, tail = head.contiguous(), tail.contiguous()
    idx_bs = Cn * dd if head.dim() == 3 and head.shape[0] > 1 else 0
    out = torch.empty(B, Cn, L * dd, dtype=torch.bfloat16, device=h0.device)
    args = _b16_args(B, Cn, S, L, dd, act, None, h0c, h0_bs, head, tail, idx_bs, out, None, trans=Ts, identity=identity)
    with _lib.on_device(h0.device):
        _lib.check(_lib.lib().recon_propagate_b16_fwd(C.byref(args), _lib.current_stream()), "recon_propagate_b16_fwd (block mode)")
    return out


class _PropagateBlocksB16(torch.autograd.Function):
    """models/models.py:240-274 on bfloat16 tensors WITH gradients and without an adjacency: the forward reads the transition tensors in
    place (and saves the states), the backward's products read them in place again, write d T_l in T's layout and sum the diagonal
    blocks of every hop into d identity (fp32, fixed order, rounded once)."""

    @staticmethod
    def forward(ctx, h0, identity, act, head, tail, n, want_states, *Ts):
        _req(h0, identity, head, tail, *Ts, dtype=torch.bfloat16)
        L, dd, B = len(Ts), identity.shape[0], Ts[0].shape[0]
        S, Cn = n * dd, n * (n - 1)
        t_shapes = [tuple(t.shape) for t in Ts]
        Ts = [t.contiguous().view(B, Cn, dd * dd) for t in Ts]
        identity, h0c = identity.contiguous(), h0.contiguous()
        h0_bs = Cn * S if h0c.dim() == 4 else 0
        head, tail = head.contiguous(), tail.contiguous()
        idx_bs = Cn * dd if head.dim() == 3 and head.shape[0] > 1 else 0
        dev = h0.device
        out = torch.empty(B, Cn, L * dd, dtype=torch.bfloat16, device=dev)
        hs = torch.empty(L, B, Cn, S, dtype=torch.bfloat16, device=dev)
        args = _b16_args(B, Cn, S, L, dd, act, None, h0c, h0_bs, head, tail, idx_bs, out, hs, trans=Ts, identity=identity)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_propagate_b16_fwd(C.byref(args), _lib.current_stream()), "recon_propagate_b16_fwd (block mode)")
        ctx.save_for_backward(h0c, identity, head, tail, hs, *Ts)
        ctx.meta = (B, Cn, S, L, dd, act, h0_bs, idx_bs, tuple(h0.shape), t_shapes)
        if want_states:
            ctx.mark_non_differentiable(hs)
            return out, hs
        return out

    @staticmethod
    def backward(ctx, gout, *_unused):
        h0c, identity, head, tail, hs, *Ts = ctx.saved_tensors
        B, Cn, S, L, dd, act, h0_bs, idx_bs, h0_shape, t_shapes = ctx.meta
        dev = gout.device
        gout = gout.contiguous()
        if gout.data_ptr() % 16:
            gout = gout.clone()
        bf = dict(dtype=torch.bfloat16, device=dev)
        g_Ts = [torch.empty(B, Cn, dd * dd, **bf) if ctx.needs_input_grad[7 + l] else None for l in range(L)]
        g_I = torch.empty(dd, dd, **bf) if ctx.needs_input_grad[1] else None
        g_h, ws = torch.empty(B, Cn, S, **bf), torch.empty(B, Cn, S, **bf)
        Lb = _lib.lib()
        fwd = _b16_args(B, Cn, S, L, dd, act, None, h0c, h0_bs, head, tail, idx_bs, gout, hs, trans=Ts, identity=identity)
        diag = torch.empty(Lb.recon_propagate_b16_bwd_diag_elems(C.byref(fwd)), **bf) if g_I is not None else None
        iws = torch.empty(Lb.recon_block_adjacency_b16_bwd_workspace_floats(dd), dtype=torch.float32, device=dev) if g_I is not None else None
        blk_idx = _index_blocks(head, tail, dd, S) if idx_bs == 0 else None
        args = _lib.PropB16BwdArgs(fwd, gout.data_ptr(), None, g_h.data_ptr(), ws.data_ptr(),
                                   _lib.ptr(blk_idx[0]) if blk_idx else None, _lib.ptr(blk_idx[1]) if blk_idx else None,
                                   _ptr_array(g_Ts), _lib.ptr(g_I), _lib.ptr(diag), _lib.ptr(iws))
        with _lib.on_device(dev):
            _lib.check(Lb.recon_propagate_b16_bwd(C.byref(args), _lib.current_stream()), "recon_propagate_b16_bwd (block mode)")
        g_h0 = None
        if ctx.needs_input_grad[0]:
            g_h0 = (g_h if h0_bs else g_h.sum(0, dtype=torch.float32).to(torch.bfloat16)).view(h0_shape)
        return (g_h0, g_I, None, None, None, None, None) + tuple(g.view(sh) if g is not None else None for g, sh in zip(g_Ts, t_shapes))


def _blocks_b16_trainable(B, n, dd, h0, T_list, identity):
    """Block mode with gradients: 2d = 16 and a state size the bfloat16 kernels take (S = 16 n: always a multiple of 8)."""
    if dd != 16 or n < 2 or B > _MAX_BATCH or B == 0 or os.environ.get("RECON_PROP_BLOCKS", "1") == "0":
        return False
    if h0.data_ptr() % 16 or identity.data_ptr() % 16 or any(T.data_ptr() % 16 for T in T_list):
        return False
    Cn, S = n * (n - 1), n * dd
    if Cn * S >= 2 ** 31 or 8 * S * 4 > 64 * 1024:
        return False
    probe = _lib.PropB16Args(B, Cn, S, len(T_list), dd, 1, None, None, 0, None, None, 0, None, None, None, None, None)
    return _lib.lib().recon_propagate_b16_form(C.byref(probe)) != 0


def propagate_blocks(T_list, identity, n, h0, nonlinearity, head_indices, tail_indices, return_states=False):
    """models/models.py:240-274 in one call: T_list = L transition tensors [B, n(n-1), (2d)^2] (or [B, n-1, n, (2d)^2]) AFTER their
    non-linearity, identity [2d, 2d]; equivalent to
        propagate([build_block_adjacency(T, identity, n) for T in T_list], h0, nonlinearity, head_indices, tail_indices)
    and computed exactly like that wherever the fused kernels do not apply (2d != 16, n > 10, batches above one launch)."""
    if nonlinearity not in _lib.ACT:
        raise NotImplementedError(nonlinearity)
    B, dd = T_list[0].shape[0], identity.shape[0]
    need_grad = torch.is_grad_enabled() and (identity.requires_grad or h0.requires_grad or any(T.requires_grad for T in T_list))
    if _float_dtype(h0, identity, *T_list) == torch.bfloat16:
        # bfloat16: the kernels read the transition tensors in place (csrc/prop_b16.hip, block mode) — inference through the fused forms,
        # training through _PropagateBlocksB16 (no adjacency is materialised in either direction); other shapes: block adjacency + propagate
        if not need_grad and _blocks_b16_available(B, n, dd, h0, T_list, identity):
            return _propagate_blocks_b16(T_list, identity, n, h0, nonlinearity, head_indices, tail_indices)
        if need_grad and _blocks_b16_trainable(B, n, dd, h0, T_list, identity):
            return _PropagateBlocksB16.apply(h0, identity, nonlinearity, head_indices, tail_indices, n, bool(return_states), *T_list)
        if return_states:
            raise NotImplementedError("return_states: the bfloat16 training form only")
        return propagate([build_block_adjacency(T, identity, n) for T in T_list], h0, nonlinearity, head_indices, tail_indices)
    if return_states:
        raise NotImplementedError("return_states: bfloat16 tensors only")
    ok = blocks_mode_available(B, n, dd, h0, need_grad, L=len(T_list), T_list=T_list)
    if not ok and need_grad and n > 10:
        ok = (blocks_mode_available(B, n, dd, h0, False, L=len(T_list), T_list=T_list) and identity.data_ptr() % 16 == 0 and
              _blocks_wide_trainable(B, n, dd, h0, len(T_list), head_indices, tail_indices))
    if not ok:
        return propagate([build_block_adjacency(T, identity, n) for T in T_list], h0, nonlinearity, head_indices, tail_indices)
    return _PropagateBlocks.apply(h0, identity, nonlinearity, head_indices, tail_indices, n, *T_list)


# ------------------------------------------------------------------------------- P4
def make_start_entity_embeddings(entity_embeddings, entity_pos_indices, unique_entities, embedding_dim,
                                 max_occurred_entity_in_batch_pos, start_embedding_template, max_num_nodes=9):
    """utils/context_utils.py:387-426, same argument list (unique_entities and the most-frequent-entity
    hint only steer a speed trick there and do not change the result).  Returns [B, C, 2dn, 1].
    Forward and backward are kernels; the gradient of `entity_embeddings` is a fixed-order segment sum (bitwise run-to-run)."""
    _req(entity_embeddings, start_embedding_template)
    n, d = max_num_nodes, embedding_dim
    B, Cn = entity_pos_indices.shape[:2]
    S = 2 * d * n
    pos = entity_pos_indices.to(device=entity_embeddings.device, dtype=torch.int64).contiguous()
    return _StartEntity.apply(entity_embeddings, pos, start_embedding_template.contiguous().view(Cn, S), B, n, d)


class _StartEntity(torch.autograd.Function):
    @staticmethod
    def forward(ctx, ent, pos, templ, B, n, d):
        ent = ent.contiguous()
        Cn, S = n * (n - 1), 2 * d * n
        out = torch.empty(B, Cn, S, 1, dtype=torch.float32, device=ent.device)
        with _lib.on_device(ent.device):
            _lib.check(_lib.lib().recon_start_entity_embeddings(ent.data_ptr(), pos.data_ptr(), templ.data_ptr(), B, n, d,
                                                                out.data_ptr(), _lib.current_stream()),
                       "recon_start_entity_embeddings")
        ctx.save_for_backward(pos, templ)
        ctx.dims = (B, n, d, ent.shape[0])
        return out

    @staticmethod
    def backward(ctx, g):
        """Fixed summation order, no torch op: one kernel cuts the two d-wide pieces per (b, c) out of g * templ, the
        SpecialSpmmFinal walk (csrc/gat.hip: k_rowsum_walk / _fix over a destination-only CSR keyed on the entity ids) adds the rows of
        each entity in slot order.  The segment key — and with it the CSR, through prepare_graph's cache — is kept per position tensor."""
        from .gat_layers import _rowsum_keyed
        pos, templ = ctx.saved_tensors
        B, n, d, U = ctx.dims
        Cn = n * (n - 1)
        dev = g.device
        g = g.contiguous()
        if g.dtype != torch.float32:
            g = g.float()
        rows = torch.empty(2 * B * Cn, d, dtype=torch.float32, device=dev)
        ck = (pos.data_ptr(), pos._version, tuple(pos.shape), str(dev))
        hit = _P4_KEYS.get(ck)
        key = None if hit is not None else torch.empty(2, 2 * B * Cn, dtype=torch.int64, device=dev)
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_start_entity_embeddings_bwd(g.data_ptr(), pos.data_ptr(), templ.data_ptr(), B, n, d, rows.data_ptr(),
                                                                    _lib.ptr(key), _lib.current_stream()), "recon_start_entity_embeddings_bwd")
        if hit is None:
            if len(_P4_KEYS) >= 4:
                _P4_KEYS.pop(next(iter(_P4_KEYS)))
            hit = _P4_KEYS[ck] = (key, pos)                               # `pos` pins data_ptr identity while cached
        return _rowsum_keyed(rows, hit[0], U), None, None, None, None, None


_P4_KEYS = {}
