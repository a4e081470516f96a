"""Drop-in replacement for the reference module `models/layers.py`: `GraphConvolution` (same
constructor, `forward(input, adj)`, parameters `weight` [in,out] / `bias` [out] with the reference's
uniform(-1/sqrt(out), 1/sqrt(out)) init) and `SparseMM`, running in csrc/prop.hip + gemm_f32.hip.

    from recon_amd.gcn_layers import GraphConvolution          # models/models.py:8

Extension over the reference: `forward` also accepts a batch, input [B,n,in] with adj [B,n,n]
(the reference's torch.mm only takes the 2-D single-graph form).  Any n is accepted (the aggregate kernel tiles
the adjacency in 32 x 32 blocks); B <= 65 535 graphs per call."""
import ctypes as C
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
        with torch.cuda.device(dev):
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
        with torch.cuda.device(dev):
            _lib.check(L.recon_gcn_bwd(C.byref(args), _lib.current_stream()), "recon_gcn_bwd")
        xs, adjs = ctx.shapes
        return (g_x.view(xs) if g_x is not None else None, g_adj.view(adjs) if g_adj is not None else None, g_w, g_b)


class SparseMM(torch.autograd.Function):
    """models/layers.py:9-32 is a legacy (non-static) autograd Function for `mm` that current torch can no
    longer run; this static equivalent keeps the name and the gradients dA = g B^T, dB = A^T g."""

    @staticmethod
    def forward(ctx, matrix1, matrix2):
        ctx.save_for_backward(matrix1, matrix2)
        return torch.mm(matrix1, matrix2)

    @staticmethod
    def backward(ctx, grad_output):
        matrix1, matrix2 = ctx.saved_tensors
        g1 = torch.mm(grad_output, matrix2.t()) if ctx.needs_input_grad[0] else None
        g2 = torch.mm(matrix1.t(), grad_output) if ctx.needs_input_grad[1] else None
        return g1, g2


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

    def forward(self, input, adj):
        return _GcnFunction.apply(input, adj, self.weight, self.bias)

    def __repr__(self):
        return self.__class__.__name__ + ' (' + str(self.in_features) + ' -> ' + str(self.out_features) + ')'
