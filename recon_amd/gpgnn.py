"""Stage-B caller of the propagation path, shaped like the reference's `GPGNN` (models/models.py:85-277):
same constructor, forward signature and state_dict keys (word_embedding.weight, pos_embedding.weight, rnn1.*,
representation_to_adj[.i].weight/bias, identity_transformation, start_embedding, head_indices, tail_indices,
linear3.*), so a checkpoint written by the reference's train.py:415-416 loads unchanged.  The sentence encoder
(embeddings, LSTM, Linear) stays stock PyTorch-ROCm (MIOpen / rocBLAS); the block-adjacency construction and the
L-hop propagation with its fused head*tail gather run on the HIP kernels of csrc/prop.hip.  SURVEY.md 8f row N3."""
import torch
import torch.nn as nn
import torch.nn.functional as F

from .propagation import (build_block_adjacency, propagate, propagate_blocks, make_start_embedding, make_start_entity_embeddings, get_head_indices,
                          get_tail_indices)


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
            Ts = [T] * L
        else:                                                            # :238-259: one per hop, `non-linear1` on it
            Ts = []
            for i in range(L):
                T = self.representation_to_adj[i](rnn_result)
                if p['non-linear1'] != "linear":
                    T = getattr(F, p['non-linear1'])(T)
                Ts.append(T)
        # :240-274 — block adjacency + propagation; fused (A_l never materialised) where the kernels allow, else build_block_adjacency + propagate
        relation = propagate_blocks(Ts, self.identity_transformation, n, self.start_embedding, p['non-linear1'], self.head_indices[0], self.tail_indices[0])
        return self.linear3(relation).view(B * self.MAX_EDGES_PER_GRAPH, -1)


class CharEmbeddings(nn.Module):
    """models/models.py:15-24 (state_dict key `embeddings.weight`)."""

    def __init__(self, vocab_size, embed_dim, drop_out_rate):
        super().__init__()
        self.embeddings = nn.Embedding(vocab_size, embed_dim, padding_idx=0)
        self.dropout = nn.Dropout(drop_out_rate)

    def forward(self, chars):
        return self.dropout(self.embeddings(chars))


class EntityEmbedding(nn.Module):
    """Entity attribute context encoder, models/models.py:26-83: every context line of an entity is a word sequence (word
    vectors + char-CNN features) run through an LSTM; the final states of all lines of one entity are convolved and max-pooled
    over the unmasked lines into one vector per entity.  Stock PyTorch-ROCm ops (MIOpen LSTM / convolution): this is the
    encoder in front of the propagation path, not the path.  Keys: word_embeddings.weight (the caller's table, shared),
    char_embeddings.embeddings.weight, lstm.*, conv1d.*, conv1d_entity.*."""

    def __init__(self, input_dim, hidden_dim, layers, is_bidirectional, drop_out_rate, entity_embed_dim, conv_filter_size,
                 entity_conv_filter_size, word_embeddings, char_embed_dim, max_word_len_entity, char_vocab, char_feature_size):
        super().__init__()
        self.input_dim, self.hidden_dim, self.layers = input_dim, hidden_dim, layers
        self.is_bidirectional, self.drop_rate = is_bidirectional, drop_out_rate
        self.word_embeddings = word_embeddings
        self.char_embeddings = CharEmbeddings(len(char_vocab), char_embed_dim, drop_out_rate)
        self.lstm = nn.LSTM(input_dim, hidden_dim, layers, batch_first=True, bidirectional=bool(is_bidirectional))
        self.conv1d = nn.Conv1d(char_embed_dim, char_feature_size, conv_filter_size)
        self.word_span = max_word_len_entity + conv_filter_size - 1      # chars one word occupies in the padded char sequence
        self.conv1d_entity = nn.Conv1d(2 * hidden_dim, entity_embed_dim, entity_conv_filter_size)

    def forward(self, words, chars, conv_mask):
        """words [U, lines, len], chars [U, lines, cfs-1 + len*(max_char+cfs-1)], conv_mask bool [U, lines-ecfs+1] (True = padding
        line) -> [U, entity_embed_dim]."""
        U, lines = words.shape[0], words.shape[1]
        words = words.reshape(U * lines, words.shape[2])
        chars = chars.reshape(U * lines, chars.shape[2])
        word_vec = self.word_embeddings(words)
        char_vec = self.char_embeddings(chars).permute(0, 2, 1)
        char_feat = torch.tanh(F.max_pool1d(self.conv1d(char_vec), self.word_span, self.word_span)).permute(0, 2, 1)
        _, (h_n, _) = self.lstm(torch.cat((word_vec, char_feat), -1))
        # last layer, both directions side by side (the reference reshapes to (layers, 2, batch, hidden): bidirectional only)
        h_n = h_n.view(self.layers, 2, U * lines, self.hidden_dim)[-1].permute(1, 0, 2).reshape(U, lines, 2 * self.hidden_dim)
        conv = self.conv1d_entity(h_n.permute(0, 2, 1))
        conv = conv.masked_fill(conv_mask.unsqueeze(1), -float('inf'))
        return conv.max(dim=2).values


class RECON_EAC(GPGNN):
    """The reference's `RECON_EAC` (models/models.py:279-487): GPGNN whose start embedding is built PER BATCH from the entity
    attribute context (`EntityEmbedding` -> `make_start_entity_embeddings`, utils/context_utils.py:387-426) instead of the
    fixed one-hot template.  Same constructor, forward signature and state_dict keys (GPGNN's plus entity_embedding_module.*).
    The per-batch start vectors, the block adjacency and the L-hop propagation run on the HIP kernels of csrc/prop.hip.

    Two places where the reference's code cannot be followed literally: its tied-projection branch (:401-445) hard-codes
    9 nodes AND ignores the context embeddings (it propagates the fixed template) — kept as is, through GPGNN.forward; and its
    head / tail index tensors bake in `batch_size` — the kernels take the [C,2d] pattern, so any batch size runs."""

    def __init__(self, p, embeddings, max_sent_len, n_out, char_vocab, MAX_EDGES_PER_GRAPH=72):
        super().__init__(p, embeddings, max_sent_len, n_out, MAX_EDGES_PER_GRAPH)
        n, d = p['max_num_nodes'], p['embedding_dim']
        self.head_indices = nn.Parameter(torch.LongTensor(get_head_indices(n, d, bs=p['batch_size'])), requires_grad=False)   # :338-342
        self.tail_indices = nn.Parameter(torch.LongTensor(get_tail_indices(n, d, bs=p['batch_size'])), requires_grad=False)
        self.entity_embedding_module = EntityEmbedding(
            p['char_embed_dim'] + embeddings.shape[1], p['hidden_dim_ent'], p['num_entEmb_layers'], p['is_bidirectional_ent'],
            p['drop_out_rate_ent'], p['entity_embed_dim'], p['conv_filter_size'], p['entity_conv_filter_size'], self.word_embedding,
            p['char_embed_dim'], p['max_char_len'], char_vocab, p['char_feature_size'])

    def forward(self, sentence_input, entity_markers, num_entities, unique_entites, entity_indices, context_words, context_chars,
                context_mask, entities_position, max_occurred_entity_in_batch_pos):
        p = self.p
        if self.tied:
            return super().forward(sentence_input, entity_markers, num_entities)
        n, L = p['max_num_nodes'], p['layer_number']
        entity_embeddings = self.entity_embedding_module(context_words, context_chars, context_mask)
        h0 = make_start_entity_embeddings(entity_embeddings, entities_position, unique_entites, p['embedding_dim'],
                                          max_occurred_entity_in_batch_pos, self.start_embedding, max_num_nodes=n)     # :365
        rnn_result = self.encode(sentence_input, entity_markers)
        B = rnn_result.size(0)
        Ts = []
        for i in range(L):                                               # :447-466
            T = self.representation_to_adj[i](rnn_result)
            if p['non-linear1'] != "linear":
                T = getattr(F, p['non-linear1'])(T)
            Ts.append(T)
        relation = propagate_blocks(Ts, self.identity_transformation, n, h0, p['non-linear1'], self.head_indices[0], self.tail_indices[0])     # :447-484
        return self.linear3(relation).view(B * self.MAX_EDGES_PER_GRAPH, -1)
