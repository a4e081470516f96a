"""Stage-B caller of the propagation path, shaped like the reference's `GPGNN` (models/models.py:85-277):
same constructor, forward signature and state_dict keys (word_embedding.weight, pos_embedding.weight, rnn1.*,
representation_to_adj[.i].weight/bias, identity_transformation, start_embedding, head_indices, tail_indices,
linear3.*), so a checkpoint written by the reference's train.py:415-416 loads unchanged.  The sentence encoder
(embeddings, LSTM, Linear) stays stock PyTorch-ROCm (MIOpen / rocBLAS); the block-adjacency construction and the
L-hop propagation with its fused head*tail gather run on the HIP kernels of csrc/prop.hip.  SURVEY.md 8f row N3."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .propagation import (build_block_adjacency, propagate, make_start_embedding, get_head_indices, get_tail_indices)


class GPGNN(nn.Module):
    def __init__(self, p, embeddings, max_sent_len, n_out, MAX_EDGES_PER_GRAPH=72):
        super().__init__()
        self.p = p
        n, d, L = p['max_num_nodes'], p['embedding_dim'], p['layer_number']
        self.MAX_EDGES_PER_GRAPH = n * (n - 1) if p.get('max_num_nodes') else MAX_EDGES_PER_GRAPH      # :93-96
        self.max_sent_len = max_sent_len
        self.word_embedding = nn.Embedding(embeddings.shape[0], embeddings.shape[1], padding_idx=0)   # :102-105
        self.word_embedding.weight.data.copy_(torch.from_numpy(embeddings))
        self.word_embedding.weight.requires_grad = False
        self.dropout1 = nn.Dropout(p=p['dropout1'])
        self.pos_embedding = nn.Embedding(4, p['position_emb'], padding_idx=0)                        # :109-110
        nn.init.orthogonal_(self.pos_embedding.weight)
        self.rnn1 = nn.LSTM(batch_first=True, input_size=embeddings.shape[1] + int(p['position_emb']),
                            hidden_size=int(p['units1']), num_layers=int(p['rnn1_layers']),
                            bidirectional=bool(p['bidirectional']))                                   # :113-119
        for parameter in self.rnn1.parameters():
            if len(parameter.size()) >= 2:
                nn.init.orthogonal_(parameter)
        self.dropout2 = nn.Dropout(p=p['dropout1'])
        self.tied = L == 1 or p['projection_style'] == 'tie'                                          # :123-132
        if self.tied:
            self.representation_to_adj = nn.Linear(p['units1'] * 2, (d * 2) ** 2)
            nn.init.xavier_uniform_(self.representation_to_adj.weight)
        else:
            self.representation_to_adj = nn.ModuleList([nn.Linear(p['units1'] * 2, (d * 2) ** 2) for _ in range(L)])
            for lin in self.representation_to_adj:
                nn.init.xavier_uniform_(lin.weight)
        self.identity_transformation = nn.Parameter(torch.eye(d * 2), requires_grad=True)             # :134-135
        self.start_embedding = nn.Parameter(torch.from_numpy(make_start_embedding(n, d)).float(), requires_grad=False)
        self.head_indices = nn.Parameter(torch.LongTensor(get_head_indices(n, d)), requires_grad=False)   # [50,C,2d] as stored
        self.tail_indices = nn.Parameter(torch.LongTensor(get_tail_indices(n, d)), requires_grad=False)
        self.linear3 = nn.Linear(d * 2 * L, n_out)                                                    # :143-146
        nn.init.xavier_uniform_(self.linear3.weight)

    def encode(self, sentence_input, entity_markers):
        """models/models.py:160-184: one LSTM pass per (sentence, ordered entity pair); returns [B, C, 2*units1]."""
        B, C = sentence_input.size(0), self.MAX_EDGES_PER_GRAPH
        expanded = torch.transpose(sentence_input.expand(C, B, self.max_sent_len), 0, 1)
        word = self.word_embedding(expanded.contiguous().view(-1, self.max_sent_len)).view(B, C, self.max_sent_len, -1)
        word = self.dropout1(word)
        pos = self.pos_embedding(entity_markers.contiguous().view(-1, self.max_sent_len)).view(B, C, self.max_sent_len, -1)
        merged = torch.cat([word, pos], dim=3)
        merged = merged.view(-1, self.max_sent_len, merged.size(-1))
        rnn_output, _ = self.rnn1(merged)
        u = self.p['units1']
        rnn_result = torch.cat([rnn_output[:, -1, :u], rnn_output[:, 0, u:]], dim=1).view(B, C, -1)
        return self.dropout2(rnn_result)

    def forward(self, sentence_input, entity_markers, num_entities=None):
        """(B, max_sent_len), (B, C, max_sent_len) -> (B*C, n_out).  Unlike the reference, any batch size works
        (its head/tail index tensors bake in 50, models/models.py:138-142; the kernels take the [C,2d] pattern)."""
        p = self.p
        n, L = p['max_num_nodes'], p['layer_number']
        rnn_result = self.encode(sentence_input, entity_markers)
        B = rnn_result.size(0)
        if self.tied:                                                    # :186-236: ONE transition tensor, `non-linear` on it
            T = self.representation_to_adj(rnn_result)
            if p['non-linear'] != "linear":
                T = getattr(F, p['non-linear'])(T)
            adjs = [build_block_adjacency(T, self.identity_transformation, n)] * L
        else:                                                            # :238-259: one per hop, `non-linear1` on it
            adjs = []
            for i in range(L):
                T = self.representation_to_adj[i](rnn_result)
                if p['non-linear1'] != "linear":
                    T = getattr(F, p['non-linear1'])(T)
                adjs.append(build_block_adjacency(T, self.identity_transformation, n))
        relation = propagate(adjs, self.start_embedding, p['non-linear1'], self.head_indices[0], self.tail_indices[0])   # :260-274
        return self.linear3(relation).view(B * self.MAX_EDGES_PER_GRAPH, -1)
