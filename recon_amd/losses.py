"""The loss of the stage-A (KB-GAT) training loop — SURVEY.md 8(f) N1, /root/reference/GAT/main.py:344-376.

`batch_gat_loss(gat_loss_func, train_indices, entity_embed, relation_embed)` has the reference's signature (its module-level
`args.valid_invalid_ratio_gat` is the keyword `valid_invalid_ratio_gat`, default 2 as in its run scripts).  With a
`torch.nn.MarginRankingLoss` (mean reduction) on GPU float32 tables the whole function is ONE launch forward (six row gathers, two L1
norms, the ranking loss and its mean) and, backward, one launch for the gradient rows plus two fixed-order segment sums into the tables
(csrc/loss.hip); anything else runs the reference's own op sequence on `gather_rows`.
"""
import ctypes as C

import torch

from . import _lib
from .graph import trusted, trust

_COUNTERS = {}


def _counter(dev):
    c = _COUNTERS.get(dev)
    if c is None:
        c = _COUNTERS[dev] = torch.zeros(1, dtype=torch.int32, device=dev)     # left zero by every call
    return c


class _TransEMarginLoss(torch.autograd.Function):
    @staticmethod
    def forward(ctx, entity_embed, relation_embed, train_indices, n_pos, reps, margin):
        dev = entity_embed.device
        ent, rel, tri = entity_embed.contiguous(), relation_embed.contiguous(), train_indices.contiguous()
        D, P = ent.shape[1], n_pos * reps
        need = ctx.needs_input_grad[0] or ctx.needs_input_grad[1]
        both = ctx.needs_input_grad[0] and ctx.needs_input_grad[1]      # (the usual case) one key tensor over both tables: one sort, one segment sum
        terms = torch.empty(P, dtype=torch.float32, device=dev)
        loss = torch.empty((), dtype=torch.float32, device=dev)
        keys = torch.empty(2, 6 * P, dtype=torch.int64, device=dev) if both else None
        ek = torch.empty(2, 4 * P, dtype=torch.int64, device=dev) if (need and not both) else None
        rk = torch.empty(2, 2 * P, dtype=torch.int64, device=dev) if (need and not both) else None
        with _lib.on_device(dev):
            if both:
                _lib.check(_lib.lib().recon_transe_margin_fwd_keys(ent.data_ptr(), rel.data_ptr(), tri.data_ptr(), n_pos, reps, D, float(margin), terms.data_ptr(),
                                                                   loss.data_ptr(), keys.data_ptr(), ent.shape[0], _lib.current_stream()),
                           "recon_transe_margin_fwd_keys")
            else:
                _lib.check(_lib.lib().recon_transe_margin_fwd(ent.data_ptr(), rel.data_ptr(), tri.data_ptr(), n_pos, reps, D, float(margin), terms.data_ptr(),
                                                              loss.data_ptr(), _lib.ptr(ek), _lib.ptr(rk), _counter(dev).data_ptr(), _lib.current_stream()),
                           "recon_transe_margin_fwd")
        if need:
            if both:
                trust(keys, bound=ent.shape[0] + rel.shape[0])           # copies of ids that were validated (or vouched for) below
                ctx.save_for_backward(ent, rel, tri, terms, keys)
            else:
                trust(ek, bound=ent.shape[0])
                trust(rk, bound=rel.shape[0])
                ctx.save_for_backward(ent, rel, tri, terms, ek, rk)
            ctx.meta = (n_pos, reps, both)
        return loss

    @staticmethod
    def backward(ctx, g):
        from .gat_layers import _rowsum_keyed
        n_pos, reps, both = ctx.meta
        if both:
            ent, rel, tri, terms, keys = ctx.saved_tensors
        else:
            ent, rel, tri, terms, ek, rk = ctx.saved_tensors
        dev, D, P = ent.device, ent.shape[1], n_pos * reps
        g = g.contiguous().to(torch.float32)
        rows = torch.empty(6 * P, D, dtype=torch.float32, device=dev)       # the entity rows' gradients [4 P, D], then the relation rows' [2 P, D]
        ge, gr = rows[:4 * P], rows[4 * P:]
        with _lib.on_device(dev):
            _lib.check(_lib.lib().recon_transe_margin_bwd(ent.data_ptr(), rel.data_ptr(), tri.data_ptr(), n_pos, reps, D, terms.data_ptr(), g.data_ptr(),
                                                          ge.data_ptr(), gr.data_ptr(), _lib.current_stream()), "recon_transe_margin_bwd")
        if both:
            g_all = _rowsum_keyed(rows, keys, ent.shape[0] + rel.shape[0])
            return g_all[:ent.shape[0]], g_all[ent.shape[0]:], None, None, None, None
        g_ent = _rowsum_keyed(ge, ek, ent.shape[0]) if ctx.needs_input_grad[0] else None
        g_rel = _rowsum_keyed(gr, rk, rel.shape[0]) if ctx.needs_input_grad[1] else None
        return g_ent, g_rel, None, None, None, None


def _validate(train_indices, n_ent, n_rel):
    """Ids out of range make the reference raise (index out of range); the kernels do not check.  One host round trip unless the producer
    vouches for the tensor (graph.trust: entity bound in `bound`, relation bound in `rel_bound`)."""
    from .graph import trust_bounds
    if trusted(train_indices):
        eb, rb = trust_bounds(train_indices)
        if (eb is None or eb <= n_ent) and (rb is None or rb <= n_rel):
            return
    if train_indices.numel() == 0:
        return
    lo, hi = torch.aminmax(train_indices, dim=0)
    lo, hi = lo.tolist(), hi.tolist()
    if min(lo) < 0 or hi[0] >= n_ent or hi[2] >= n_ent or hi[1] >= n_rel:
        raise IndexError("recon_amd.batch_gat_loss: triple ids out of range (entities %d, relations %d)" % (n_ent, n_rel))


def batch_gat_loss(gat_loss_func, train_indices, entity_embed, relation_embed, valid_invalid_ratio_gat=2):
    """GAT/main.py:344-376.  train_indices int64 [T, 3]: the positive triples, then 2 * valid_invalid_ratio_gat corrupted copies of them."""
    ratio = int(valid_invalid_ratio_gat)
    reps = 2 * ratio
    n_pos = int(train_indices.shape[0] / (reps + 1))
    if entity_embed.is_cuda and train_indices.device != entity_embed.device:
        train_indices = train_indices.to(entity_embed.device)            # the reference indexes a CUDA table with a CPU LongTensor (GAT/main.py:344-376)
    fused = (isinstance(gat_loss_func, torch.nn.MarginRankingLoss) and gat_loss_func.reduction == "mean" and entity_embed.is_cuda and
             relation_embed.is_cuda and relation_embed.device == entity_embed.device and
             entity_embed.dtype == torch.float32 and relation_embed.dtype == torch.float32 and train_indices.dtype == torch.int64 and
             train_indices.dim() == 2 and train_indices.shape[1] == 3 and n_pos > 0 and train_indices.shape[0] == n_pos * (reps + 1) and
             entity_embed.shape[1] == relation_embed.shape[1])
    if fused:
        _validate(train_indices, entity_embed.shape[0], relation_embed.shape[0])
        return _TransEMarginLoss.apply(entity_embed, relation_embed, train_indices, n_pos, reps, float(gat_loss_func.margin))
    # the reference's op sequence (any loss function, any dtype): rows through gather_rows where the tables are GPU float32
    from .gat_layers import gather_rows
    def rows(t, i):                                                      # chosen per table: each may live elsewhere / in another dtype
        if t.is_cuda and t.dtype == torch.float32:
            return gather_rows(t, i.to(t.device).contiguous())
        return t[i.to(t.device)]
    pos = train_indices[:n_pos].repeat(reps, 1)
    neg = train_indices[n_pos:]
    pos_norm = torch.norm(rows(entity_embed, pos[:, 0]) + rows(relation_embed, pos[:, 1]) - rows(entity_embed, pos[:, 2]), p=1, dim=1)
    neg_norm = torch.norm(rows(entity_embed, neg[:, 0]) + rows(relation_embed, neg[:, 1]) - rows(entity_embed, neg[:, 2]), p=1, dim=1)
    y = -torch.ones(reps * n_pos, device=entity_embed.device)
    loss = gat_loss_func(pos_norm, neg_norm, y)
    return loss
