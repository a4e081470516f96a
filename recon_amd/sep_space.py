"""The `GAT_sep_space` variant of the stage-A model (reference GAT_sep_space/models.py:91-245): `SpKBGATModified` with one more
trainable tensor, `W_ent2rel` [num_relation, D, D] (D = entity_out_dim_1 * nheads), that carries an entity embedding into the space
of a triple's relation before the translation loss (`tanh(e W_r)`, GAT_sep_space/main.py:359-364, :372-377) and before ConvKB
(GAT_sep_space/models.py:316-320).  Encoder, forward / batch_test signatures and every other state_dict key are the GAT tree's
(recon_amd/models.py); a checkpoint written by GAT_sep_space/main.py loads with strict=True.

    from recon_amd.sep_space import SpKBGATModified          # instead of `from models import SpKBGATModified` in GAT_sep_space/main.py
"""
import torch
import torch.nn as nn

from . import _lib
from . import models as _models
from .graph import trusted


class SpKBGATModified(_models.SpKBGATModified):
    def __init__(self, initial_entity_emb, initial_relation_emb, entity_out_dim, relation_out_dim, drop_GAT, alpha, nheads_GAT,
                 initial_entity_emb_params=None):
        super().__init__(initial_entity_emb, initial_relation_emb, entity_out_dim, relation_out_dim, drop_GAT, alpha, nheads_GAT,
                         initial_entity_emb_params)
        wide = self.entity_out_dim_1 * self.nheads_GAT_1
        self.W_ent2rel = nn.Parameter(torch.zeros(size=(self.num_relation, wide, wide)))                 # :136-138
        nn.init.xavier_uniform_(self.W_ent2rel.data, gain=1.414)
        self.nonlinearity_ent2rel = torch.tanh                                                           # :139

    def ent2rel(self, entity_rows, relation_ids):
        """`nonlinearity_ent2rel(bmm(entity_rows.unsqueeze(1), W_ent2rel[relation_ids])).squeeze(1)` (GAT_sep_space/main.py:360-364)
        without the [T, D, D] gather the reference materialises (160 KB per triple at D = 200): rows are walked in relation order, rows of
        one relation share the passes over its matrix (csrc/rel_mm.hip: one launch forward, two backward, no host read)."""
        return self.nonlinearity_ent2rel(rel_rows_mm(entity_rows, relation_ids, self.W_ent2rel))


class _RelRowsMM(torch.autograd.Function):
    """out[t] = x[t] . W[rel[t]]   (x [T, D], W [R, D, Dout], rel [T] int64 in [0, R))"""

    @staticmethod
    def forward(ctx, x, rel, W):
        T, D = x.shape
        R, _, Dout = W.shape
        xc, Wc, relc = x.contiguous(), W.contiguous(), rel.contiguous()
        out = torch.empty(T, Dout, dtype=torch.float32, device=x.device)
        order = torch.argsort(relc, stable=True).to(torch.int32)
        if T:
            with _lib.on_device(x.device):
                _lib.check(_lib.lib().recon_rel_rows_mm(xc.data_ptr(), order.data_ptr(), relc.data_ptr(), Wc.data_ptr(), T, D, Dout, 0, out.data_ptr(),
                                                        _lib.current_stream()), "recon_rel_rows_mm")
        ctx.save_for_backward(xc, relc, Wc, order)
        return out

    @staticmethod
    def backward(ctx, g):
        xc, relc, Wc, order = ctx.saved_tensors
        T, D = xc.shape
        R, _, Dout = Wc.shape
        g = g.contiguous()
        g_x = g_W = None
        with _lib.on_device(g.device):
            if ctx.needs_input_grad[0]:
                g_x = torch.empty_like(xc)
                if T:
                    _lib.check(_lib.lib().recon_rel_rows_mm(g.data_ptr(), order.data_ptr(), relc.data_ptr(), Wc.data_ptr(), T, Dout, D, 1, g_x.data_ptr(),
                                                            _lib.current_stream()), "recon_rel_rows_mm")
            if ctx.needs_input_grad[2]:
                if not T:
                    return g_x, None, torch.zeros_like(Wc)
                g_W = torch.empty_like(Wc)
                seg = torch.zeros(R + 1, dtype=torch.int32, device=g.device)
                seg[1:] = torch.cumsum(torch.bincount(relc, minlength=R), 0)
                _lib.check(_lib.lib().recon_rel_rows_mm_wgrad(xc.data_ptr(), g.data_ptr(), order.data_ptr(), seg.data_ptr(), R, D, Dout, g_W.data_ptr(),
                                                              _lib.current_stream()), "recon_rel_rows_mm_wgrad")
        return g_x, None, g_W


def rel_rows_mm(x, rel, W):
    """Per-row product with the matrix its relation selects: `torch.bmm(x.unsqueeze(1), W[rel]).squeeze(1)` (GAT_sep_space/main.py:359-364)
    on the device kernels of csrc/rel_mm.hip.  float32, x [T, D], W [R, D, Dout] with D <= 1024, Dout <= 512; relation ids outside [0, R)
    raise (asynchronously, like an out-of-range index on the device does in the reference)."""
    if not x.is_cuda:
        raise RuntimeError("rel_rows_mm needs device tensors (there is no CPU path)")
    if x.dtype != torch.float32 or W.dtype != torch.float32 or rel.dtype != torch.int64:
        raise TypeError("rel_rows_mm takes float32 rows / matrices and int64 relation ids")
    if x.dim() != 2 or W.dim() != 3 or rel.shape != (x.shape[0],) or W.shape[1] != x.shape[1]:
        raise ValueError("rel_rows_mm: x [T, D], rel [T], W [R, D, Dout]")
    if x.shape[1] > 1024 or W.shape[2] > 512:
        raise ValueError("rel_rows_mm: D <= 1024 and Dout <= 512")
    if rel.numel() and not trusted(rel, limit=W.shape[0]):
        torch._assert_async(((rel >= 0) & (rel < W.shape[0])).all())
    return _RelRowsMM.apply(x, rel, W)
