"""The `GAT_sep_space` variant of the stage-A model (reference GAT_sep_space/models.py:91-245): `SpKBGATModified` with one more
trainable tensor, `W_ent2rel` [num_relation, D, D] (D = entity_out_dim_1 * nheads), that carries an entity embedding into the space
of a triple's relation before the translation loss (`tanh(e W_r)`, GAT_sep_space/main.py:359-364, :372-377) and before ConvKB
(GAT_sep_space/models.py:316-320).  Encoder, forward / batch_test signatures and every other state_dict key are the GAT tree's
(recon_amd/models.py); a checkpoint written by GAT_sep_space/main.py loads with strict=True.

    from recon_amd.sep_space import SpKBGATModified          # instead of `from models import SpKBGATModified` in GAT_sep_space/main.py
"""
import torch
import torch.nn as nn

from . import models as _models
from .gat_layers import small_mm


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
        without the [T, D, D] gather the reference materialises (160 KB per triple at D = 200): triples are grouped by relation
        (stable sort), each group is one [T_r, D] x [D, D] product on the library-free GEMM, rows return to the caller's order."""
        W = self.W_ent2rel
        T, D = entity_rows.shape
        out = entity_rows.new_empty(T, W.shape[2])
        if T == 0:
            return out
        order = torch.argsort(relation_ids, stable=True)
        counts = torch.bincount(relation_ids, minlength=W.shape[0]).tolist()            # one host read per call: group extents
        rows = entity_rows.index_select(0, order)
        pieces, lo = [], 0
        for r, c in enumerate(counts):
            if c:
                pieces.append(small_mm(rows[lo:lo + c].contiguous(), W[r]))
                lo += c
        grouped = torch.cat(pieces, 0)
        return self.nonlinearity_ent2rel(out.index_copy(0, order, grouped))
