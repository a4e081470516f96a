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

    def relation_features(self, sentence_input, entity_markers, context_words, context_chars, context_mask, entities_position,
                          max_occurred_entity_in_batch_pos):
        """Untied branch up to the classifier (:365-484): [B, C, 2d L] head*tail features of every hop."""
        p = self.p
        n, L = p['max_num_nodes'], p['layer_number']
        entity_embeddings = self.entity_embedding_module(context_words, context_chars, context_mask)
        h0 = make_start_entity_embeddings(entity_embeddings, entities_position, None, p['embedding_dim'],
                                          max_occurred_entity_in_batch_pos, self.start_embedding, max_num_nodes=n)     # :365
        rnn_result = self.encode(sentence_input, entity_markers)
        Ts = []
        for i in range(L):                                               # :447-466
            T = self.representation_to_adj[i](rnn_result)
            if p['non-linear1'] != "linear":
                T = getattr(F, p['non-linear1'])(T)
            Ts.append(T)
        return propagate_blocks(Ts, self.identity_transformation, n, h0, p['non-linear1'], self.head_indices[0], self.tail_indices[0])     # :447-484

    def forward(self, sentence_input, entity_markers, num_entities, unique_entites, entity_indices, context_words, context_chars,
                context_mask, entities_position, max_occurred_entity_in_batch_pos):
        if self.tied:
            return GPGNN.forward(self, sentence_input, entity_markers, num_entities)
        relation = self.relation_features(sentence_input, entity_markers, context_words, context_chars, context_mask, entities_position,
                                          max_occurred_entity_in_batch_pos)
        return self.linear3(relation).view(relation.size(0) * self.MAX_EDGES_PER_GRAPH, -1)


class RECON_EAC_KGGAT(RECON_EAC):
    """The reference's `RECON_EAC_KGGAT` (models/models.py:489-701): RECON_EAC whose classifier also sees the stage-A KB-GAT
    embeddings of each pair's head and tail entity (`gat_entity_embeddings` [B, C, 2 * gat_entity_embedding_dim], concatenated
    behind the propagation features, :693-697).  Same constructor, forward signature and state_dict keys as the reference.

    The reference's tied-projection branch (:597-649) feeds the propagation features alone to a classifier sized for the
    concatenation and fails in `linear3`; here it fails with the same exception type before any work is done."""

    def __init__(self, p, embeddings, max_sent_len, n_out, char_vocab, MAX_EDGES_PER_GRAPH=72):
        super().__init__(p, embeddings, max_sent_len, n_out, char_vocab, MAX_EDGES_PER_GRAPH)
        self.linear3 = nn.Linear(p['embedding_dim'] * 2 * p['layer_number'] + 2 * p['gat_entity_embedding_dim'], n_out)      # :547-549
        nn.init.xavier_uniform_(self.linear3.weight)

    def forward(self, sentence_input, entity_markers, num_entities, unique_entites, entity_indices, context_words, context_chars,
                context_mask, entities_position, max_occurred_entity_in_batch_pos, gat_entity_embeddings):
        if self.tied:
            raise RuntimeError("RECON_EAC_KGGAT: the tied projection feeds %d features to a classifier built for %d (models/models.py:622, :647)"
                               % (self.p['embedding_dim'] * 2 * self.p['layer_number'], self.linear3.in_features))
        relation = self.relation_features(sentence_input, entity_markers, context_words, context_chars, context_mask, entities_position,
                                          max_occurred_entity_in_batch_pos)
        both = torch.cat([relation, gat_entity_embeddings.to(relation.dtype)], dim=-1)                                      # :693-694
        return self.linear3(both).view(relation.size(0) * self.MAX_EDGES_PER_GRAPH, -1)


class RECON(RECON_EAC):
    """The reference's full model `RECON` (models/models.py:703-968): RECON_EAC + the KB-GAT entity embeddings of every pair +
    a per-relation triple score in the relation space of GAT_sep_space: for each pair with known embeddings,
    `|| tanh(head W_r) + g_r - tanh(tail W_r) ||_1` for every output relation r (:940-958), scattered into a [B*C, n_out]
    block behind the other features.  Constructor arguments, forward signature and state_dict keys follow the reference
    (`gat_relation_embeddings` trainable, `W_ent2rel` frozen; head / tail index tensors are plain attributes there, :759-768,
    and non-persistent buffers here so that `.to(device)` moves them and checkpoints stay key-compatible).

    `gat_relation_embeddings` is the dict the reference loads from JSON (string index -> vector), `W_ent2rel_all_rels` the
    [n_gat_rel, ent_dim, rel_dim] array, `idx2property` output index -> property, `gat_relation2idx` property -> string index."""

    def __init__(self, p, embeddings, max_sent_len, n_out, char_vocab, gat_relation_embeddings, W_ent2rel_all_rels, idx2property,
                 gat_relation2idx, MAX_EDGES_PER_GRAPH=72):
        super().__init__(p, embeddings, max_sent_len, n_out, char_vocab, MAX_EDGES_PER_GRAPH)
        n, d, L = p['max_num_nodes'], p['embedding_dim'], p['layer_number']
        head, tail = self.head_indices.data, self.tail_indices.data
        del self.head_indices, self.tail_indices
        self.register_buffer("head_indices", head, persistent=False)
        self.register_buffer("tail_indices", tail, persistent=False)
        self.linear3 = nn.Linear(d * 2 * L + 2 * p['gat_entity_embedding_dim'] + n_out, n_out)                               # :772-773
        nn.init.xavier_uniform_(self.linear3.weight)
        W_all = torch.as_tensor(W_ent2rel_all_rels, dtype=torch.float32)
        rel0 = torch.zeros(n_out, len(gat_relation_embeddings["0"]))                                                        # :779-785
        W0 = torch.zeros(n_out, W_all.shape[1], W_all.shape[2])                                                             # :787-793
        for i in range(n_out):
            gat_idx = gat_relation2idx.get(idx2property[i], None)
            if gat_idx is not None:
                rel0[i] = torch.as_tensor(gat_relation_embeddings[gat_idx], dtype=torch.float32)
                W0[i] = W_all[int(gat_idx)]
        self.gat_relation_embeddings = nn.Parameter(rel0, requires_grad=True)
        self.W_ent2rel = nn.Parameter(W0, requires_grad=False)

    def translation_scores(self, nonzero_gat_entity_embeddings):
        """[M, 2 ent_dim] (head | tail) -> [M, n_out] L1 translation residuals in each output relation's space (:934-953).
        All relations at once: one [M, ent_dim] x [ent_dim, n_out * rel_dim] product per side on the library-free GEMM."""
        from .gat_layers import small_mm
        W = self.W_ent2rel
        n_out, ent_dim, rel_dim = W.shape
        half = nonzero_gat_entity_embeddings.shape[-1] // 2
        emb = nonzero_gat_entity_embeddings.to(device=W.device, dtype=W.dtype)
        Wf = W.permute(1, 0, 2).reshape(ent_dim, n_out * rel_dim)
        head = torch.tanh(small_mm(emb[:, :half].contiguous(), Wf)).view(-1, n_out, rel_dim)
        tail = torch.tanh(small_mm(emb[:, half:].contiguous(), Wf)).view(-1, n_out, rel_dim)
        return (head + self.gat_relation_embeddings.unsqueeze(0) - tail).abs().sum(-1)

    def forward(self, sentence_input, entity_markers, num_entities, unique_entites, entity_indices, context_words, context_chars,
                context_mask, entities_position, max_occurred_entity_in_batch_pos, nonzero_gat_entity_embeddings, nonzero_entity_pos,
                gat_entity_embeddings):
        if self.tied:
            raise RuntimeError("RECON: the tied projection feeds %d features to a classifier built for %d (models/models.py:836, :861)"
                               % (self.p['embedding_dim'] * 2 * self.p['layer_number'], self.linear3.in_features))
        relation = self.relation_features(sentence_input, entity_markers, context_words, context_chars, context_mask, entities_position,
                                          max_occurred_entity_in_batch_pos)
        rows = relation.size(0) * self.MAX_EDGES_PER_GRAPH
        relation = relation.reshape(rows, -1)                                                                               # :926
        gat = gat_entity_embeddings.to(relation.dtype).reshape(rows, -1)                                                    # :927
        scores = torch.zeros(rows, self.W_ent2rel.shape[0], device=relation.device, dtype=relation.dtype)                  # :954-958
        if nonzero_gat_entity_embeddings.shape[0] > 0:
            scores = scores.index_put((nonzero_entity_pos.to(relation.device),), self.translation_scores(nonzero_gat_entity_embeddings))
        return self.linear3(torch.cat([relation, gat, scores], dim=-1))                                                     # :960-961
