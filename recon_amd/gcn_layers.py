"""Drop-in replacement for the reference module `models/layers.py`: `GraphConvolution` (same
constructor, `forward(input, adj)`, parameters `weight` [in,out] / `bias` [out] with the reference's
uniform(-1/sqrt(out), 1/sqrt(out)) init) and `SparseMM`, running in csrc/prop.hip + gemm_f32.hip.

    from recon_amd.gcn_layers import GraphConvolution          # models/models.py:8

bfloat16 tensors (`layer.to(torch.bfloat16)`, bf16 input / adj) take the bf16 MFMA path of csrc/gemm_b16.hip / gcn_b16.hip:
bf16 storage, fp32 accumulation, what torch.mm does for bf16 operands.

Extension over the reference: `forward` also accepts a batch, input [B,n,in] with adj [B,n,n]
(the reference's torch.mm only takes the 2-D single-graph form).  Any n is accepted (the aggregate kernel tiles
the adjacency in 32 x 32 blocks); B <= 65 535 graphs per call."""
import ctypes as C
import os
import math

import torch
from torch.nn.parameter import Parameter
from torch.nn.modules.module import Module

from . import _lib
from . import gat_layers as _gl        # the GEMM-family switch (_GEMM_BX3) is ONE module-level setting for GAT and GCN


def _req(*ts):
    for t in ts:
        if t is None:
            continue
        if not t.is_cuda:
            raise RuntimeError("recon_amd: expected a GPU tensor (this package has no CPU path)")
        if t.dtype != torch.float32:
            raise TypeError("recon_amd: the HIP kernels compute in float32, got %s" % t.dtype)


class _GcnFunction(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, adj, weight, bias):
        _req(x, adj, weight, bias)
        x3 = x.contiguous().view(-1, x.shape[-2], x.shape[-1])
        adj3 = adj.contiguous().view(-1, adj.shape[-2], adj.shape[-1])
        weight = weight.contiguous()
        B, n, I = x3.shape
        O = weight.shape[1]
        if adj3.shape != (B, n, n) or weight.shape[0] != I:
            raise ValueError("GraphConvolution: inconsistent shapes")
        dev = x.device
        sup = torch.empty(B, n, O, dtype=torch.float32, device=dev)
        # allocated in the caller's shape and returned AS IS: the tensor saved for the backward (its sign is the ReLU mask)
        # is the tensor the caller holds, so an in-place edit of the result is caught by autograd's version check
        out = torch.empty(x.shape[:-1] + (O,), dtype=torch.float32, device=dev)
        w_split = None
        mode = _gl._GEMM_BX3                                  # split-precision GEMMs (fp32-accurate, csrc/gemm_bx3.hip): auto | 1 | 0
        if mode == "1" or (mode != "0" and 2.0 * B * n * I * O >= 2.0e9):    # small products do not pay for the term-plane launches
            w_split = torch.empty(_lib.lib().recon_gcn_split_bytes(I, O), dtype=torch.uint8, device=dev)
        args = _lib.GcnArgs(B, n, I, O, x3.data_ptr(), adj3.data_ptr(), weight.data_ptr(), _lib.ptr(bias),
                            sup.data_ptr(), out.data_ptr(), _lib.ptr(w_split))
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_gcn_fwd(C.byref(args), _lib.current_stream()), "recon_gcn_fwd")
        ctx.save_for_backward(x3, adj3, weight, bias, sup, out, w_split)
        ctx.shapes = (tuple(x.shape), tuple(adj.shape))
        return out

    @staticmethod
    def backward(ctx, gout):
        x3, adj3, weight, bias, sup, out, w_split = ctx.saved_tensors
        B, n, I = x3.shape
        O = weight.shape[1]
        dev = gout.device
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=dev)
        gout = gout.contiguous()
        nx, nadj, nw, nb = ctx.needs_input_grad
        g_sup = torch.empty(B, n, O, **f32)
        partial = torch.empty(L.recon_gcn_bwd_partial_floats(B, n, I, O), **f32)
        g_x = torch.empty(B, n, I, **f32) if nx else None
        g_adj = torch.empty(B, n, n, **f32) if nadj else None
        g_w = torch.empty(I, O, **f32) if nw else None
        g_b = torch.empty(O, **f32) if (nb and bias is not None) else None
        fwd = _lib.GcnArgs(B, n, I, O, x3.data_ptr(), adj3.data_ptr(), weight.data_ptr(), _lib.ptr(bias), sup.data_ptr(),
                           out.data_ptr(), _lib.ptr(w_split))
        gs_split = (torch.empty(L.recon_gcn_bwd_split_bytes(B, n, O), dtype=torch.uint8, device=dev)
                    if (w_split is not None and g_w is not None) else None)
        args = _lib.GcnBwdArgs(fwd, gout.data_ptr(), g_sup.data_ptr(), partial.data_ptr(), _lib.ptr(g_x), _lib.ptr(g_adj),
                               _lib.ptr(g_w), _lib.ptr(g_b), _lib.ptr(gs_split))
        with _lib.on_device(dev):
            _lib.check(L.recon_gcn_bwd(C.byref(args), _lib.current_stream()), "recon_gcn_bwd")
        xs, adjs = ctx.shapes
        return (g_x.view(xs) if g_x is not None else None, g_adj.view(adjs) if g_adj is not None else None, g_w, g_b)


def _rows_view(t, feat, pads_read=True):
    """(data_ptr-compatible 2-D view info) of a [..., n, feat] bf16 tensor whose rows are feat contiguous elements at a regular
    row stride ld with ld % 8 == 0 and ld >= feat rounded up to 8 — the layout the bf16 kernels read in place (a contiguous tensor
    with feat % 8 == 0, or the padded views this module hands out).  Returns ld, or None if the tensor has to be repacked."""
    if t.dim() < 2 or t.stride(-1) != 1:
        return None
    if pads_read and feat % 8 and not getattr(t, "_recon_padded", False):
        # the kernels read the pad columns feat .. round_up(feat, 8) of every row (times zero weights): only rows THIS module wrote have
        # zeros there — somebody else's `buf[..., :feat]` may carry inf / NaN in them, and 0 * NaN would poison the result
        return None
    ld = t.stride(-2)
    if ld % 8 or ld < (feat + 7) // 8 * 8 or t.data_ptr() % 16:
        return None
    rows = t.shape[-2]
    for d in range(t.dim() - 3, -1, -1):                       # leading dims must continue the same row sequence
        if t.shape[d] != 1 and t.stride(d) != rows * ld:
            return None
        rows *= t.shape[d]
    return ld


def _rows_in_place(t, feat):
    """Row stride of a [..., n, feat] bf16 tensor a kernel with masked row tails can read where it lies: rows of feat contiguous elements
    at a regular EVEN stride >= feat, 4-byte aligned.  None: repack."""
    if t.dim() < 2 or t.stride(-1) != 1:
        return None
    ld = t.stride(-2)
    if ld % 2 or ld < feat or t.data_ptr() % 4:
        return None
    rows = t.shape[-2]
    for d in range(t.dim() - 3, -1, -1):
        if t.shape[d] != 1 and t.stride(d) != rows * ld:
            return None
        rows *= t.shape[d]
    return ld


def _packed_rows(t, feat, zero_pad):
    """[rows, ld] bf16 buffer holding t's rows (ld = feat rounded up to 8); pad columns zeroed when the kernels multiply them."""
    ld = (feat + 7) // 8 * 8
    t2 = t.reshape(-1, feat)
    if ld == feat:
        return t2.contiguous(), ld
    buf = (torch.zeros if zero_pad else torch.empty)(t2.shape[0], ld, dtype=torch.bfloat16, device=t.device)
    buf[:, :feat] = t2
    return buf, ld


_ZEROS = {}
_FUSED = os.environ.get("RECON_GCN_FUSED", "1") != "0"


def _zero_page(dev):
    z = _ZEROS.get(dev)
    if z is None:
        z = _ZEROS[dev] = torch.zeros(1024, dtype=torch.uint8, device=dev)
    return z


class _NoGradCtx:
    """Stands in for autograd's ctx when a forward runs outside autograd (torch.no_grad())."""
    needs_input_grad = (False, False, False, False, False)

    def save_for_backward(self, *tensors):
        pass


_NO_GRAD_CTX = _NoGradCtx()
_PLANES_CACHE = os.environ.get("RECON_GCN_PLANES_CACHE", "1") != "0"
_PLANES = {}        # (weight data_ptr, version, in, out, device) -> (planes, weight): repacked weights of frozen layers


class _GcnB16Function(torch.autograd.Function):
    """The layer on bfloat16 tensors (BASELINE.json configs[2]): bf16 storage, fp32 accumulation, the three feature products on
    the bf16 matrix cores (csrc/gemm_b16.hip, gcn_b16.hip).  Feature counts that are not multiples of 8 live in row-padded
    buffers; the result is returned as a [..., :out] view of one, which the next layer reads in place."""

    @staticmethod
    def forward(ctx, x, adj, weight, bias, keep_planes=False):
        for t in (x, adj, weight, bias):
            if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16):
                raise TypeError("recon_amd: the bfloat16 GraphConvolution path needs bfloat16 GPU tensors for input, adj, weight and bias")
        n, I = x.shape[-2], x.shape[-1]
        O = weight.shape[1]
        B = x.numel() // (n * I) if x.numel() else 0
        adj3 = adj.contiguous().view(-1, n, n) if adj.numel() else adj.reshape(0, n, n)
        if adj3.shape[0] != B or weight.shape[0] != I:
            raise ValueError("GraphConvolution: inconsistent shapes")
        dev = x.device
        L = _lib.lib()
        o8 = (O + 7) // 8 * 8
        bf = dict(dtype=torch.bfloat16, device=dev)
        # `support` = x @ W is only needed again for d adj; without it the forward is ONE kernel (csrc/gcn_b16.hip k_gcn_b16_fused_fwd:
        # n <= 32, out <= 320) that keeps it in registers
        fused = (not ctx.needs_input_grad[1]) and n <= 32 and o8 <= 320 and B * n * max((I + 7) // 8 * 8, o8) * 2 < 2 ** 31 - 1 and _FUSED
        # inference through the fused kernel: it masks the K tail, so what lies behind a row's last feature is never multiplied — anybody's
        # padded rows, and unpadded ones (4-byte aligned), are read in place; training keeps rows it can hand to the weight-gradient GEMM
        masked = fused and not any(ctx.needs_input_grad)
        ldx = _rows_view(x, I, pads_read=not masked)
        xr = x
        if ldx is None and masked and I % 2 == 0 and x.is_contiguous() and x.data_ptr() % 16 == 0:
            ldx = I
        elif ldx is None:
            xr, ldx = _packed_rows(x, I, zero_pad=True)
        sup = None if fused else torch.empty(B * n, o8, **bf)
        out_p = torch.empty(B * n, o8, **bf)
        weight = weight.contiguous()
        # W^T / W repacked for the matrix cores: the weight of a module in eval() mode keeps its planes across calls (`keep_planes`), keyed
        # on identity + version (an in-place update through the tensor bumps the version; one through `.data` does NOT — the module drops
        # its planes in reset_parameters / load_state_dict / _apply and offers invalidate_planes()); otherwise repacked every call
        key = (weight.data_ptr(), weight._version, I, O, str(dev))
        frozen = _PLANES_CACHE and keep_planes and not ctx.needs_input_grad[2]
        hit = _PLANES.get(key) if frozen else None
        planes = hit[0] if hit is not None else torch.empty(L.recon_gcn_b16_planes_bytes(I, O), dtype=torch.uint8, device=dev)
        args = _lib.GcnB16Args(B, n, I, O, xr.data_ptr(), ldx, adj3.data_ptr(), weight.data_ptr(), _lib.ptr(bias), _lib.ptr(sup), o8,
                               out_p.data_ptr(), o8, planes.data_ptr(), 1 if hit is not None else 0, None, None, 0)
        with _lib.on_device(dev):
            _lib.check(L.recon_gcn_b16_fwd(C.byref(args), _lib.current_stream()), "recon_gcn_b16_fwd")
        if hit is None and frozen:
            if len(_PLANES) >= 64:
                _PLANES.clear()
            _PLANES[key] = (planes, weight)                              # keeps `weight` alive: its data_ptr is the key
        ctx.save_for_backward(xr, adj3, weight, bias, sup, out_p, planes)
        ctx.meta = (B, n, I, O, ldx, o8, tuple(x.shape), tuple(adj.shape))
        if o8 == O:
            return out_p.view(x.shape[:-1] + (O,))
        if not fused:                                                     # the next layer (and the backward) read these columns in place;
            out_p[:, O:].zero_()                                          # the fused kernel writes them as zeros itself
        out = out_p.as_strided(x.shape[:-1] + (O,), _strides(x.shape[:-1], o8))
        out._recon_padded = True
        return out

    @staticmethod
    def backward(ctx, gout):
        xr, adj3, weight, bias, sup, out_p, planes = ctx.saved_tensors
        B, n, I, O, ldx, o8, xs, adjs = ctx.meta
        dev = gout.device
        L = _lib.lib()
        bf = dict(dtype=torch.bfloat16, device=dev)
        nx, nadj, nw, nb = ctx.needs_input_grad[:4]
        if gout.dtype != torch.bfloat16:
            gout = gout.to(torch.bfloat16)
        ldg = _rows_view(gout, O, pads_read=False)                 # pad columns of a gradient are never read
        gr = gout
        if ldg is None:
            gr, ldg = _packed_rows(gout, O, zero_pad=False)     # pad columns of a gradient are never read
        i8 = (I + 7) // 8 * 8
        g_sup = torch.empty(B * n, o8, **bf)
        partial = torch.empty(L.recon_gcn_b16_bwd_partial_floats(B, n, I, O), dtype=torch.float32, device=dev)
        g_x = torch.empty(B * n, i8, **bf) if nx else None
        g_adj = torch.empty(B, n, n, **bf) if nadj else None
        g_w = torch.empty(I, O, **bf) if nw else None
        g_b = torch.empty(O, **bf) if (nb and bias is not None) else None
        fwd = _lib.GcnB16Args(B, n, I, O, xr.data_ptr(), ldx, adj3.data_ptr(), weight.data_ptr(), _lib.ptr(bias), _lib.ptr(sup), o8,
                              out_p.data_ptr(), o8, planes.data_ptr(), 1, None, None, 0)
        args = _lib.GcnB16BwdArgs(fwd, gr.data_ptr(), ldg, g_sup.data_ptr(), partial.data_ptr(), _lib.ptr(g_x), i8, _lib.ptr(g_adj),
                                  _lib.ptr(g_w), _lib.ptr(g_b), _zero_page(dev).data_ptr())
        with _lib.on_device(dev):
            _lib.check(L.recon_gcn_b16_bwd(C.byref(args), _lib.current_stream()), "recon_gcn_b16_bwd")
        if g_x is not None:
            if i8 == I:
                g_x = g_x.view(xs)
            else:                       # pad columns stay unwritten and the view untagged: nothing reads the pad columns of a gradient
                g_x = g_x.as_strided(xs, _strides(xs[:-1], i8))   # (_rows_view(..., pads_read=False) in the producer layer's backward)
        return g_x, (g_adj.view(adjs) if g_adj is not None else None), g_w, g_b, None


class RaggedAdjacency:
    """A batch of graphs of DIFFERENT sizes for GraphConvolution (BASELINE.json configs[4]: power-law graphs of up to 256 nodes each):
    graph b owns node rows node_ptr[b] .. node_ptr[b+1] of the layer's input [N, in] and a dense n_b x n_b adjacency stored at
    values[adj_ptr[b] : adj_ptr[b] + n_b^2] (row major).  Equivalent to the reference layer (models/layers.py:57-63) applied to each graph,
    i.e. to one block-diagonal adjacency over all N nodes.  `values` is a bfloat16 GPU tensor and may require grad."""

    def __init__(self, values, sizes):
        sizes = [int(v) for v in sizes]
        if any(v <= 0 for v in sizes):
            raise ValueError("RaggedAdjacency: every graph needs at least one node")
        self.sizes = sizes
        self.B = len(sizes)
        self.n_max = max(sizes) if sizes else 0
        self.total_rows = sum(sizes)
        if values.dim() != 1 or values.numel() != sum(v * v for v in sizes):
            raise ValueError("RaggedAdjacency: `values` must be the flat concatenation of the graphs' n_b x n_b adjacencies")
        self.values = values
        npt = torch.tensor([0] + sizes, dtype=torch.int64).cumsum(0)
        apt = torch.tensor([0] + [v * v for v in sizes], dtype=torch.int64).cumsum(0)
        self.node_ptr = npt.to(dtype=torch.int32, device=values.device)
        self.adj_ptr = apt.to(device=values.device)

    @classmethod
    def from_dense(cls, mats):
        """From a list of [n_b, n_b] tensors."""
        return cls(torch.cat([m.reshape(-1) for m in mats]), [m.shape[0] for m in mats])

    def block(self, b):
        o, n = int(self.adj_ptr[b]), self.sizes[b]
        return self.values[o:o + n * n].view(n, n)


class _GcnB16RaggedFunction(torch.autograd.Function):
    """GraphConvolution over a ragged batch in bfloat16: x @ W over all node rows at once (bf16 matrix cores), the aggregate per graph."""

    @staticmethod
    def forward(ctx, x, values, weight, bias, ragged):
        for t in (x, values, weight, bias):
            if t is not None and (not t.is_cuda or t.dtype != torch.bfloat16):
                raise TypeError("recon_amd: the ragged GraphConvolution path needs bfloat16 GPU tensors for input, adjacency values, weight and bias")
        if x.dim() != 2 or x.shape[0] != ragged.total_rows or weight.shape[0] != x.shape[1]:
            raise ValueError("GraphConvolution (ragged): input must be [total nodes, in_features]")
        if ragged.B > _MAX_BATCH:
            raise NotImplementedError("GraphConvolution (ragged): more than %d graphs per call" % _MAX_BATCH)
        N, I = x.shape
        O = weight.shape[1]
        dev = x.device
        L = _lib.lib()
        o8 = (O + 7) // 8 * 8
        bf = dict(dtype=torch.bfloat16, device=dev)
        ldx = _rows_view(x, I)
        xr = x
        if ldx is None:
            xr, ldx = _packed_rows(x, I, zero_pad=True)
        values = values.contiguous()
        sup = torch.empty(N, o8, **bf)
        out_p = torch.empty(N, o8, **bf)
        weight = weight.contiguous()
        planes = torch.empty(L.recon_gcn_b16_planes_bytes(I, O), dtype=torch.uint8, device=dev)
        args = _lib.GcnB16Args(ragged.B, ragged.n_max, I, O, xr.data_ptr(), ldx, values.data_ptr(), weight.data_ptr(), _lib.ptr(bias), sup.data_ptr(), o8,
                               out_p.data_ptr(), o8, planes.data_ptr(), 0, ragged.node_ptr.data_ptr(), ragged.adj_ptr.data_ptr(), N)
        with _lib.on_device(dev):
            _lib.check(L.recon_gcn_b16_fwd(C.byref(args), _lib.current_stream()), "recon_gcn_b16_fwd (ragged)")
        ctx.save_for_backward(xr, values, weight, bias, sup, out_p, planes)
        ctx.ragged = ragged
        ctx.meta = (N, I, O, ldx, o8)
        if o8 == O:
            return out_p
        out_p[:, O:].zero_()
        out = out_p.as_strided((N, O), (o8, 1))
        out._recon_padded = True
        return out

    @staticmethod
    def backward(ctx, gout):
        xr, values, weight, bias, sup, out_p, planes = ctx.saved_tensors
        N, I, O, ldx, o8 = ctx.meta
        rg = ctx.ragged
        dev = gout.device
        L = _lib.lib()
        bf = dict(dtype=torch.bfloat16, device=dev)
        nx, nadj, nw, nb, _ = ctx.needs_input_grad
        if gout.dtype != torch.bfloat16:
            gout = gout.to(torch.bfloat16)
        ldg = _rows_view(gout, O, pads_read=False)
        gr = gout
        if ldg is None:
            gr, ldg = _packed_rows(gout, O, zero_pad=False)
        i8 = (I + 7) // 8 * 8
        g_sup = torch.empty(N, o8, **bf)
        partial = torch.empty(L.recon_gcn_b16_bwd_partial_floats(1, N, I, O) + rg.B * O, dtype=torch.float32, device=dev)
        g_x = torch.empty(N, i8, **bf) if nx else None
        g_adj = torch.empty_like(values) if nadj else None
        g_w = torch.empty(I, O, **bf) if nw else None
        g_b = torch.empty(O, **bf) if (nb and bias is not None) else None
        fwd = _lib.GcnB16Args(rg.B, rg.n_max, I, O, xr.data_ptr(), ldx, values.data_ptr(), weight.data_ptr(), _lib.ptr(bias), sup.data_ptr(), o8,
                              out_p.data_ptr(), o8, planes.data_ptr(), 1, rg.node_ptr.data_ptr(), rg.adj_ptr.data_ptr(), N)
        args = _lib.GcnB16BwdArgs(fwd, gr.data_ptr(), ldg, g_sup.data_ptr(), partial.data_ptr(), _lib.ptr(g_x), i8, _lib.ptr(g_adj),
                                  _lib.ptr(g_w), _lib.ptr(g_b), _zero_page(dev).data_ptr())
        with _lib.on_device(dev):
            _lib.check(L.recon_gcn_b16_bwd(C.byref(args), _lib.current_stream()), "recon_gcn_b16_bwd (ragged)")
        if g_x is not None and i8 != I:
            g_x = g_x.as_strided((N, I), (i8, 1))
        return g_x, g_adj, g_w, g_b, None


def _strides(lead, ld):
    """Strides of a [*lead, feat] view over rows of stride ld (lead = leading dims ending with the row dim)."""
    st, acc = [], ld
    for d in reversed(lead):
        st.append(acc)
        acc *= d
    return tuple(reversed(st)) + (1,)


_MAX_BATCH = 65535          # graphs per launch (recon_gcn_fwd/bwd return RECON_ERR_UNSUPPORTED above)


class SparseMM(torch.autograd.Function):
    """models/layers.py:9-32 is a legacy (non-static) autograd Function for `mm` that current torch can no
    longer run; this static equivalent keeps the name and the gradients dA = g B^T, dB = A^T g."""

    @staticmethod
    def forward(ctx, matrix1, matrix2):
        ctx.save_for_backward(matrix1, matrix2)
        return _mm(matrix1, matrix2, False, False)

    @staticmethod
    def backward(ctx, grad_output):
        matrix1, matrix2 = ctx.saved_tensors
        g1 = _mm(grad_output, matrix2, False, True) if ctx.needs_input_grad[0] else None
        g2 = _mm(matrix1, grad_output, True, False) if ctx.needs_input_grad[1] else None
        return g1, g2


def _mm(a, b, a_t, b_t):
    """op(a) op(b): fp32 GPU matrices on this library's fp32 matrix-core GEMM (recon_sgemm_ex); anything else (the class is kept for
    name compatibility and is also importable on a CPU-only host) on torch.mm."""
    if a.is_cuda and b.is_cuda and a.dtype == torch.float32 and b.dtype == torch.float32 and a.dim() == 2 and b.dim() == 2:
        from .gat_layers import _sgemm_ex
        return _sgemm_ex(a.contiguous(), a_t, b.contiguous(), b_t)
    return torch.mm(a.t() if a_t else a, b.t() if b_t else b)


_STACK_PLANES = {}     # (weight data_ptr, version, in, out, device) -> (W^T planes, weight)


def invalidate_stack_planes():
    """Drop the repacked weights gcn_stack() keeps (after an in-place write through `weight.data`)."""
    _STACK_PLANES.clear()


def _ptr_array(tensors):
    return (C.c_void_p * len(tensors))(*[(t.data_ptr() if t is not None else None) for t in tensors])


def _ptrs(base, offsets):
    return (C.c_void_p * len(offsets))(*[base + o for o in offsets])


class _GcnB16StackFunction(torch.autograd.Function):
    """`for l in layers: x = l(x, adj)` with gradients, as one launch forward and three launches backward (csrc/gcn_b16.hip:
    k_gcn_b16_stack_fwd keeping every layer's result; k_gcn_b16_stack_bwd; one split-K launch + one second pass for all weight and bias
    gradients): models/layers.py:57-63 applied L times.  Output, g_x and g_bias are bit-equal to the loop over _GcnB16Function, g_W up to
    the summation order of its split-K partials.  `adj` gets no gradient here (gcn_stack() routes such calls through the loop).
    Host side: one allocation per KIND of buffer (results, repacked weights, g_support, gradients), not per layer — at cfg 3a the loop's
    ~40 small allocations and launches were as long as its kernels."""

    @staticmethod
    def forward(ctx, x, adj, n_layers, *params):
        ws, bs = list(params[0::2]), list(params[1::2])
        n, I = x.shape[-2], x.shape[-1]
        D = ws[0].shape[1]
        B = x.numel() // (n * I)
        dev = x.device
        L = _lib.lib()
        o8 = (D + 7) // 8 * 8
        # rows the weight-gradient GEMM of layer 0 can read: 16-byte aligned, stride % 8 == 0 (its pad columns only reach rows of the product
        # that are never stored, so they need no zeros; the forward kernel masks its own K tail)
        ldx = _rows_view(x, I, pads_read=False)
        xr, xin, ldin = x, x, ldx
        if ldx is None:
            ldin = _rows_in_place(x, I)
            if ldin is not None and x.data_ptr() % 16 == 0:          # the forward kernel reads x where it lies and leaves the aligned copy itself
                ldx = (I + 7) // 8 * 8
                xr = torch.empty(B * n, ldx, dtype=torch.bfloat16, device=dev)
            else:
                xr, ldx = _packed_rows(x, I, zero_pad=False)
                xin, ldin = xr, ldx
        adj3 = adj.contiguous().view(-1, n, n)
        ws = [w.contiguous() for w in ws]
        acts = torch.empty(n_layers, B * n, o8, dtype=torch.bfloat16, device=dev)
        pb = [(L.recon_gcn_b16_planes_bytes(w.shape[0], D) + 255) // 256 * 256 for w in ws]
        poff = [sum(pb[:l]) for l in range(n_layers)]
        planes = torch.empty(sum(pb), dtype=torch.uint8, device=dev)
        args = _lib.GcnB16StackTrainArgs(B, n, I, D, n_layers, xin.data_ptr(), ldin, xr.data_ptr() if xr is not xin else None, ldx, adj3.data_ptr(),
                                         _ptr_array(ws), _ptr_array(bs),
                                         _ptrs(planes.data_ptr(), poff), _ptrs(acts.data_ptr(), [l * B * n * o8 * 2 for l in range(n_layers)]), o8)
        with _lib.on_device(dev):
            _lib.check(L.recon_gcn_b16_stack_train_fwd(C.byref(args), _lib.current_stream()), "recon_gcn_b16_stack_train_fwd")
        ctx.save_for_backward(xr, adj3, acts, planes, *ws, *[b for b in bs if b is not None])
        ctx.meta = (B, n, I, D, n_layers, ldx, o8, tuple(x.shape), [b is not None for b in bs], poff)
        out_p = acts[n_layers - 1]
        if o8 == D:
            return out_p.view(x.shape[:-1] + (D,))
        out = out_p.as_strided(x.shape[:-1] + (D,), _strides(x.shape[:-1], o8), out_p.storage_offset())
        out._recon_padded = True
        return out

    @staticmethod
    def backward(ctx, gout):
        B, n, I, D, nl, ldx, o8, xs, has_b, poff = ctx.meta
        sv = ctx.saved_tensors
        xr, adj3, acts, planes = sv[0], sv[1], sv[2], sv[3]
        ws = sv[4:4 + nl]
        bs, it = [], iter(sv[4 + nl:])
        for h in has_b:
            bs.append(next(it) if h else None)
        dev = gout.device
        L = _lib.lib()
        bf = dict(dtype=torch.bfloat16, device=dev)
        if gout.dtype != torch.bfloat16:
            gout = gout.to(torch.bfloat16)
        ldg = _rows_in_place(gout, D)                               # read in place through masked loads: any even row stride
        gr = gout
        if ldg is None:
            gr, ldg = _packed_rows(gout, D, zero_pad=False)
        i8 = I if I % 4 == 0 else (I + 7) // 8 * 8                  # g_x leaves the kernel as 8-byte stores: a dense [.., I] result where I % 4 == 0
        need = ctx.needs_input_grad
        rows = B * n
        g_sup = torch.empty(nl, rows, o8, **bf)
        partial = torch.empty(L.recon_gcn_b16_stack_bwd_partial_floats(B, n, I, D, nl), dtype=torch.float32, device=dev)
        g_x = torch.empty(rows, i8, **bf) if need[0] else None
        # the parameters' gradients: one buffer [g_W_0 | g_W_1 .. | g_b_0 ..], handed back as views
        wn = [w.shape[0] * D for w in ws]
        gbuf = torch.empty(sum(wn) + nl * D, **bf)
        g_w = [gbuf[sum(wn[:l]):sum(wn[:l + 1])].view(ws[l].shape[0], D) if need[3 + 2 * l] else None for l in range(nl)]
        g_b = [gbuf[sum(wn) + l * D:sum(wn) + (l + 1) * D] if (bs[l] is not None and need[4 + 2 * l]) else None for l in range(nl)]
        args = _lib.GcnB16StackTrainArgs(B, n, I, D, nl, xr.data_ptr(), ldx, None, 0, adj3.data_ptr(), _ptr_array(ws), _ptr_array(bs),
                                         _ptrs(planes.data_ptr(), poff), _ptrs(acts.data_ptr(), [l * rows * o8 * 2 for l in range(nl)]), o8,
                                         gr.data_ptr(), ldg, _ptrs(g_sup.data_ptr(), [l * rows * o8 * 2 for l in range(nl)]), partial.data_ptr(),
                                         _lib.ptr(g_x), i8, _ptr_array(g_w), _ptr_array(g_b), _zero_page(dev).data_ptr())
        with _lib.on_device(dev):
            _lib.check(L.recon_gcn_b16_stack_train_bwd(C.byref(args), _lib.current_stream()), "recon_gcn_b16_stack_train_bwd")
        if g_x is not None:
            g_x = g_x.view(xs) if i8 == I else g_x.as_strided(xs, _strides(xs[:-1], i8))
        grads = []
        for l in range(nl):
            grads += [g_w[l], g_b[l]]
        return (g_x, None, None, *grads)


def gcn_stack(x, adj, layers):
    """`for l in layers: x = l(x, adj)` — the reference's stacks of GraphConvolutions over one adjacency (models/layers.py:57-63 applied
    len(layers) times) — as ONE launch where that is possible: bfloat16 tensors, graphs of n <= 32 nodes with n % 4 == 0, every layer
    hidden -> hidden after the first with hidden % 4 == 0, hidden <= 320.  The activations of a graph stay in LDS between the layers
    (csrc/gcn_b16.hip: k_gcn_b16_stack_fwd); with gradients (anything but `adj` requiring grad, in_features <= 320) the forward also keeps
    every layer's result and the backward is the mirror-image kernel + one weight-gradient GEMM per layer (_GcnB16StackFunction).  Results
    and gradients are bit-equal to the loop, which is also what runs for every other case.  Repacked weights are kept per weight tensor (identity + version); call
    invalidate_stack_planes() after writing through `weight.data`."""
    layers = list(layers)
    if isinstance(adj, RaggedAdjacency):                                    # graphs of different sizes: layer by layer (GEMM over all rows + aggregate per graph;
        for l in layers:                                                    # the one-kernel form of a layer was measured and is no faster there)
            x = l(x, adj)
        return x
    need_grad = torch.is_grad_enabled() and (x.requires_grad or adj.requires_grad or any(p.requires_grad for l in layers for p in l.parameters()))
    ok = (len(layers) >= 2 and len(layers) <= 8 and x.is_cuda and x.dtype == torch.bfloat16 and adj.dtype == torch.bfloat16
          and x.dim() in (2, 3) and os.environ.get("RECON_GCN_STACK", "1") != "0" and not (need_grad and adj.requires_grad))
    if ok:
        n, I = x.shape[-2], x.shape[-1]
        B = x.numel() // (n * I) if x.numel() else 0
        D = layers[0].out_features
        ok = (B > 0 and n <= 32 and n % 4 == 0 and D % 4 == 0 and (D + 7) // 8 * 8 <= 320 and layers[0].in_features == I
              and all(l.in_features == D and l.out_features == D for l in layers[1:])
              and all(l.weight.dtype == torch.bfloat16 and (l.bias is None or l.bias.data_ptr() % 8 == 0) for l in layers)
              and adj.is_contiguous() and adj.data_ptr() % 8 == 0
              and adj.numel() == B * n * n and B * n * max((I + 7) // 8 * 8, (D + 7) // 8 * 8) * 2 < 2 ** 31 - 1 and B <= 4 * _MAX_BATCH)
        if ok and need_grad:
            ok = (I + 7) // 8 * 8 <= 320 and os.environ.get("RECON_GCN_STACK_TRAIN", "1") != "0"
        elif ok:
            ok = x.is_contiguous() and I % 2 == 0 and x.data_ptr() % 16 == 0
    if ok and need_grad:
        params = []
        for l in layers:
            params += [l.weight, l.bias]
        return _GcnB16StackFunction.apply(x, adj, len(layers), *params)
    if not ok:
        for l in layers:
            x = l(x, adj)
        return x
    dev = x.device
    L = _lib.lib()
    o8 = (D + 7) // 8 * 8
    planes = []
    with _lib.on_device(dev):
        for l in layers:
            w = l.weight
            key = (w.data_ptr(), w._version, l.in_features, D, str(dev))
            hit = _STACK_PLANES.get(key)
            if hit is None:
                kp = (l.in_features + 31) // 32 * 32
                pl = torch.empty(D * kp, dtype=torch.bfloat16, device=dev)
                _lib.check(L.recon_gcn_b16_transposed_planes(w.detach().contiguous().data_ptr(), l.in_features, D, pl.data_ptr(), _lib.current_stream()),
                           "recon_gcn_b16_transposed_planes")
                if len(_STACK_PLANES) >= 64:
                    _STACK_PLANES.clear()
                hit = _STACK_PLANES[key] = (pl, w)
            planes.append(hit[0])
        out_p = torch.empty(B * n, o8, dtype=torch.bfloat16, device=dev)
        parr = (C.c_void_p * len(layers))(*[pl.data_ptr() for pl in planes])
        barr = (C.c_void_p * len(layers))(*[(l.bias.data_ptr() if l.bias is not None else None) for l in layers])
        args = _lib.GcnB16StackArgs(B, n, I, D, len(layers), x.data_ptr(), I, adj.data_ptr(), parr, barr, out_p.data_ptr(), o8)
        rc = L.recon_gcn_b16_stack_fwd(C.byref(args), _lib.current_stream())
    if rc == -2:                                                            # RECON_ERR_UNSUPPORTED: layer by layer
        for l in layers:
            x = l(x, adj)
        return x
    _lib.check(rc, "recon_gcn_b16_stack_fwd")
    if o8 == D:
        return out_p.view(x.shape[:-1] + (D,))
    out = out_p.as_strided(x.shape[:-1] + (D,), _strides(x.shape[:-1], o8))
    out._recon_padded = True
    return out


class GraphConvolution(Module):
    """Simple GCN layer, models/layers.py:35-68: relu(adj @ (input @ weight) + bias)."""

    def __init__(self, in_features, out_features, bias=True):
        super().__init__()
        self.in_features = in_features
        self.out_features = out_features
        self.weight = Parameter(torch.Tensor(in_features, out_features))
        if bias:
            self.bias = Parameter(torch.Tensor(out_features))
        else:
            self.register_parameter('bias', None)
        self.reset_parameters()

    def reset_parameters(self):
        stdv = 1. / math.sqrt(self.weight.size(1))
        self.weight.data.uniform_(-stdv, stdv)
        if self.bias is not None:
            self.bias.data.uniform_(-stdv, stdv)
        self.invalidate_planes()

    def invalidate_planes(self):
        """Drop the repacked copies of `weight` an eval()-mode bfloat16 layer keeps across calls.  Needed after an in-place write through
        `weight.data` (which autograd's version counter does not see); reset_parameters, load_state_dict and .to() / .cuda() call it."""
        for k in [k for k, v in _PLANES.items() if v[1] is getattr(self, "weight", None) or k[0] == self.weight.data_ptr()]:
            _PLANES.pop(k, None)

    def _load_from_state_dict(self, *args, **kwargs):
        super()._load_from_state_dict(*args, **kwargs)
        self.invalidate_planes()

    def _apply(self, fn, *args, **kwargs):
        self.invalidate_planes()
        return super()._apply(fn, *args, **kwargs)

    def forward(self, input, adj):
        if isinstance(adj, RaggedAdjacency):                  # graphs of different sizes (bfloat16): input [total nodes, in]
            return _GcnB16RaggedFunction.apply(input, adj.values, self.weight, self.bias, adj)
        fn = _GcnB16Function if input.dtype == torch.bfloat16 else _GcnFunction     # bf16 storage / fp32 accumulate (module.to(torch.bfloat16))
        extra = (not self.training,) if fn is _GcnB16Function else ()              # eval(): the repacked weight planes are kept across calls
        if input.dim() == 3 and input.shape[0] > _MAX_BATCH:   # 16-bit grid dimension over the graphs; graphs are independent: run slices
            B = input.shape[0]
            return torch.cat([fn.apply(input[b0:b0 + _MAX_BATCH], adj[b0:b0 + _MAX_BATCH] if adj.dim() == 3 else adj, self.weight, self.bias, *extra)
                              for b0 in range(0, B, _MAX_BATCH)], dim=0)
        if fn is _GcnB16Function and not torch.is_grad_enabled():
            # inference: nothing is recorded, so the autograd.Function machinery (~10 us per call, a third of this layer's host time at
            # cfg 3a) is skipped; the same forward runs with a context that saves nothing
            return _GcnB16Function.forward(_NO_GRAD_CTX, input, adj, self.weight, self.bias, *extra)
        return fn.apply(input, adj, self.weight, self.bias, *extra)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
