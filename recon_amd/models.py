"""Callers of the hot path, shaped like the reference's model classes so that checkpoints and call
sites carry over.  `SpGAT` mirrors GAT/models.py:11-88 (same constructor, forward signature and
state_dict keys: attention_i.a / attention_i.a_2 / W / out_att.a / out_att.a_2) but runs all heads
in ONE fused launch instead of a Python loop over H layers."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .gat_layers import SpGraphAttentionLayer, gat_heads, cat_edge_embed, gather_rows
from .graph import prepare_graph


class SpGAT(nn.Module):
    def __init__(self, num_nodes, nfeat, nhid, relation_dim, dropout, alpha, nheads):
        super().__init__()
        self.dropout = dropout
        self.dropout_layer = nn.Dropout(self.dropout)
        self.attentions = [SpGraphAttentionLayer(num_nodes, nfeat, nhid, relation_dim, dropout=dropout,
                                                 alpha=alpha, concat=True) for _ in range(nheads)]
        for i, attention in enumerate(self.attentions):
            self.add_module('attention_{}'.format(i), attention)
        self.W = nn.Parameter(torch.zeros(size=(relation_dim, nheads * nhid)))
        nn.init.xavier_uniform_(self.W.data, gain=1.414)
        self.out_att = SpGraphAttentionLayer(num_nodes, nhid * nheads, nheads * nhid, nheads * nhid,
                                             dropout=dropout, alpha=alpha, concat=False)
        self.alpha = alpha

    def heads_forward(self, x, edge_list, edge_embed, edge_list_nhop, edge_embed_nhop):
        """The H-head attention stage (GAT/models.py:71-72) as one fused call."""
        graph = prepare_graph(edge_list, edge_list_nhop, x.shape[0])
        ee = cat_edge_embed(edge_embed, edge_list_nhop, edge_embed_nhop)
        a = torch.stack([att.a for att in self.attentions])               # [H, D, 2F+R]
        a2 = torch.cat([att.a_2 for att in self.attentions], dim=0)       # [H, D]
        keeps = [att.draw_keep(graph.E, x.device) for att in self.attentions]   # reference draw order
        keep = torch.cat(keeps, dim=0) if keeps[0] is not None else None
        return gat_heads(x, ee, a, a2, graph, keep, self.alpha, True)

    def forward(self, Corpus_, entity_embeddings, relation_embed, edge_list, edge_type, edge_embed,
                edge_list_nhop, edge_type_nhop):
        x = entity_embeddings
        has_nhop = edge_type_nhop.shape[0] != 0
        if has_nhop:
            edge_embed_nhop = gather_rows(relation_embed, edge_type_nhop[:, 0]) + gather_rows(relation_embed, edge_type_nhop[:, 1])
        else:
            edge_embed_nhop = torch.tensor([])
        x = self.heads_forward(x, edge_list, edge_embed, edge_list_nhop, edge_embed_nhop)
        x = self.dropout_layer(x)
        out_relation_1 = relation_embed.mm(self.W)
        edge_embed = gather_rows(out_relation_1, edge_type)
        if has_nhop:
            edge_embed_nhop = gather_rows(out_relation_1, edge_type_nhop[:, 0]) + gather_rows(out_relation_1, edge_type_nhop[:, 1])
        else:
            edge_embed_nhop = torch.tensor([])
        x = F.elu(self.out_att(x, edge_list, edge_embed, edge_list_nhop, edge_embed_nhop))
        return x, out_relation_1
