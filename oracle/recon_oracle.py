"""CPU ORACLE for the RECON graph-context aggregation hot path.  TEST INFRASTRUCTURE ONLY.

This module restates, in plain torch-on-CPU tensor algebra, what the reference computes on the
path named by BASELINE.json:north_star.  It is the CHECKER: only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.  Nothing under
`recon_amd/` imports it, and the product path has no CPU fallback.

Parity status: PINNED.  The reference has no golden vectors of its own for this path
(SURVEY.md section 4), so the oracle is pinned against outputs of the reference itself, run
in the build container by `tests/golden/gen_golden.py` and committed as `tests/golden/*.npz`;
`tests/test_oracle_golden.py` checks every function below against them (fp32: 1e-5 abs,
fp64: 1e-12).

All `file:line` citations are relative to the reference tree (ansonb/RECON).
Floating point throughout; works for float32 and float64 inputs.
"""
import itertools

import numpy as np
import torch
import torch.nn.functional as F


# =============================================================================== GAT side
def spmm_rowsum(edge, edge_w, N):
    """out[r,:] = sum over edges e with edge[0,e]==r of edge_w[e,:]   (GAT/layers.py:54-64).

    The reference builds a hybrid COO [N,N,out] tensor and sums over dim 1; the column index
    never enters the result and duplicate (row, col) pairs are summed.
    """
    out = torch.zeros(N, edge_w.shape[1], dtype=edge_w.dtype)
    out.index_add_(0, edge[0], edge_w)
    return out


def spmm_rowsum_backward(edge, grad_out):
    """grad_edge_w = grad_out[edge[0]]   (GAT/layers.py:67-79)."""
    return grad_out[edge[0]]


def spmm_rowsum_aten_sequence(edge, edge_w, N):
    """Same result through the op sequence the reference itself issues (sparse_coo_tensor ->
    torch.sparse.sum(dim=1) -> to_dense, GAT/layers.py:56-64).  Used for the timed CPU baseline
    so that its cost profile (coalesce + sort + copies) is representative of the reference."""
    a = torch.sparse_coo_tensor(edge, edge_w, torch.Size([N, N, edge_w.shape[1]]))
    return torch.sparse.sum(a, dim=1).to_dense()


def _cat_edges(edge, edge_embed, edge_list_nhop, edge_embed_nhop):
    # GAT/layers.py:124-127 : n-hop edges are appended when present; an absent n-hop arrives
    # as a float tensor of shape [0] (GAT/models.py:57,81).
    if edge_list_nhop is not None and edge_list_nhop.shape[0] > 0:
        edge = torch.cat((edge, edge_list_nhop), dim=1)
        edge_embed = torch.cat((edge_embed, edge_embed_nhop), dim=0)
    return edge, edge_embed


def gat_layer_forward(x, edge, edge_embed, edge_list_nhop, edge_embed_nhop, a, a_2, alpha, concat,
                      mask=None, aten_sequence=False, return_intermediates=False):
    """SpGraphAttentionLayer.forward   (GAT/layers.py:111-178).

    m_e = a . [x[dst_e]; x[src_e]; r_e]          (:129-137)
    w_e = exp(-leakyrelu_alpha(a_2 . m_e))       (:143-146)   no max subtraction
    Z_i = sum_{e: dst_e = i} w_e ; Z_i == 0 -> 1e-12            (:150-152)
    w_e <- dropout(w_e)  (AFTER the row sum; `mask` holds the 0 / 1/(1-p) factors)   (:158)
    h_i = (sum_e w_e m_e) / Z_i                  (:161-169)
    out = elu(h) if concat else h                (:173-178)
    """
    N = x.shape[0]
    edge, edge_embed = _cat_edges(edge, edge_embed, edge_list_nhop, edge_embed_nhop)
    dst, src = edge[0], edge[1]
    edge_h = torch.cat((x[dst], x[src], edge_embed), dim=1)          # [E, 2F+R]
    m = edge_h @ a.t()                                                # [E, D]
    sigma = (m @ a_2.t()).squeeze(1)                                  # [E]
    w = torch.exp(-F.leaky_relu(sigma, alpha))
    rowsum = spmm_rowsum_aten_sequence if aten_sequence else spmm_rowsum
    Z = rowsum(edge, w.unsqueeze(1), N)
    Z = torch.where(Z == 0.0, torch.full_like(Z, 1e-12), Z)
    wk = w if mask is None else w * mask.to(w.dtype)
    U = rowsum(edge, wk.unsqueeze(1) * m, N)
    h = U / Z
    out = F.elu(h) if concat else h
    if return_intermediates:
        return out, dict(m=m, sigma=sigma, w=w, Z=Z.squeeze(1), U=U, h=h, edge=edge, edge_embed=edge_embed)
    return out


def gat_layer_backward(x, edge, edge_embed, edge_list_nhop, edge_embed_nhop, a, a_2, alpha, concat,
                       grad_out, mask=None):
    """Closed-form gradients of gat_layer_forward (what autograd derives
    from GAT/layers.py:111-178 together with the custom backward at :67-79).

    Returns dict with g_x, g_edge_embed (rows of the concatenated 1-hop + n-hop list), g_a, g_a_2
    and the per-edge / per-node intermediates the HIP kernels are checked against stage by stage.
    """
    N, Fdim = x.shape
    out, im = gat_layer_forward(x, edge, edge_embed, edge_list_nhop, edge_embed_nhop, a, a_2, alpha,
                                concat, mask=mask, return_intermediates=True)
    edge, ee = im["edge"], im["edge_embed"]
    dst, src = edge[0], edge[1]
    m, sigma, w, Z, h = im["m"], im["sigma"], im["w"], im["Z"], im["h"]
    k = torch.ones_like(w) if mask is None else mask.to(w.dtype)
    g_h = grad_out * torch.where(h > 0, torch.ones_like(h), torch.exp(h)) if concat else grad_out
    gU = g_h / Z[:, None]
    gZ = -(g_h * h).sum(1) / Z
    t = (gU[dst] * m).sum(1)
    gw = k * t + gZ[dst]
    gsig = -gw * w * torch.where(sigma > 0, torch.ones_like(sigma), torch.full_like(sigma, alpha))
    gm = (k * w)[:, None] * gU[dst] + gsig[:, None] * a_2[0][None, :]
    g_a2 = (gsig[:, None] * m).sum(0, keepdim=True)
    gP_dst = torch.zeros(N, m.shape[1], dtype=m.dtype).index_add_(0, dst, gm)
    gP_src = torch.zeros(N, m.shape[1], dtype=m.dtype).index_add_(0, src, gm)
    A_dst, A_src, A_rel = a[:, :Fdim], a[:, Fdim:2 * Fdim], a[:, 2 * Fdim:]
    g_x = gP_dst @ A_dst + gP_src @ A_src
    g_ee = gm @ A_rel
    g_a = torch.cat((gP_dst.t() @ x, gP_src.t() @ x, gm.t() @ ee), dim=1)
    return dict(out=out, g_x=g_x, g_edge_embed=g_ee, g_a=g_a, g_a_2=g_a2, gm=gm, gP_dst=gP_dst,
                gP_src=gP_src, sigma=sigma, w=w, Z=Z, m=m, g_sigma=gsig)


def spgat_forward(x, relation_embed, edge_list, edge_type, edge_embed, edge_list_nhop, edge_type_nhop,
                  head_a, head_a2, W, out_a, out_a2, alpha, masks=None, aten_sequence=False, layer_mask=None, out_mask=None):
    """SpGAT.forward   (GAT/models.py:47-88).  Eval mode / dropout 0 by default; train mode with the dropout factors handed in, in the
    order the reference draws them: `masks[h]` — head h's E-vector (GAT/layers.py:158, drawn inside the list comprehension of :71-72),
    `layer_mask` — dropout_layer on the concatenated heads [N, H*D] (:73), `out_mask` — out_att's E-vector (:86).

    head_a / head_a2: lists of per-head parameters (attention_i.a, attention_i.a_2).
    Returns (x_out [N, H*D], out_relation_1 [n_rel, H*D]).
    """
    has_nhop = edge_type_nhop is not None and edge_type_nhop.shape[0] > 0
    ee_nhop = (relation_embed[edge_type_nhop[:, 0]] + relation_embed[edge_type_nhop[:, 1]]) if has_nhop else None
    nhop = edge_list_nhop if has_nhop else None
    heads = [gat_layer_forward(x, edge_list, edge_embed, nhop, ee_nhop, a, a2, alpha, True,
                               mask=None if masks is None else masks[i], aten_sequence=aten_sequence)
             for i, (a, a2) in enumerate(zip(head_a, head_a2))]
    xc = torch.cat(heads, dim=1)                                       # :71-72
    if layer_mask is not None:
        xc = xc * layer_mask.to(xc.dtype)                              # :73
    out_rel = relation_embed @ W                                       # :77
    ee = out_rel[edge_type]                                            # :79
    ee_nhop = (out_rel[edge_type_nhop[:, 0]] + out_rel[edge_type_nhop[:, 1]]) if has_nhop else None
    y = F.elu(gat_layer_forward(xc, edge_list, ee, nhop, ee_nhop, out_a, out_a2, alpha, False,
                                mask=out_mask, aten_sequence=aten_sequence))          # :86-87
    return y, out_rel


def nhop_edges(train_indices_nhop):
    """GAT/models.py:145-148: 2-hop quadruples [E2,4] = (source, rel_1, rel_2, target) ->
    edge_list_nhop [2,E2] = (target, source), edge_type_nhop [E2,2] = (rel_1, rel_2)."""
    if train_indices_nhop is None or train_indices_nhop.shape[0] == 0:
        return None, None
    return torch.stack((train_indices_nhop[:, 3], train_indices_nhop[:, 0])), train_indices_nhop[:, 1:3]


def spkbgat_forward(entity_embeddings, relation_embeddings, batch_entities, edge_list, edge_type, train_indices_nhop,
                    head_a, head_a2, W, out_a, out_a2, W_entities, alpha, masks=None, layer_mask=None, out_mask=None):
    """SpKBGATModified.forward / batch_test (GAT/models.py:136-185, 188-239); eval mode unless the dropout factors are handed in
    (spgat_forward).
    `entity_embeddings` is the table AFTER the in-place L2 normalisation of :160 (the reference normalises
    `.data`, so no gradient flows through that normalisation).  Returns (out_entity, out_relation, mask)."""
    nhop_list, nhop_type = nhop_edges(train_indices_nhop)
    x, out_rel = spgat_forward(entity_embeddings, relation_embeddings, edge_list, edge_type, relation_embeddings[edge_type],
                               nhop_list, nhop_type, head_a, head_a2, W, out_a, out_a2, alpha, masks=masks, layer_mask=layer_mask,
                               out_mask=out_mask)
    mask = torch.zeros(entity_embeddings.shape[0], dtype=entity_embeddings.dtype)
    mask[torch.unique(batch_entities)] = 1.0                                    # :167-177
    out = entity_embeddings @ W_entities + mask[:, None] * x                    # :179-181
    return F.normalize(out, p=2, dim=1), out_rel, mask                          # :183


# ------------------------------------------------------------------------------- stage-A batch builders
def batch_gat_loss(train_indices, entity_embed, relation_embed, valid_invalid_ratio_gat=2, margin=1.0):
    """GAT/main.py:344-376 with gat_loss_func = nn.MarginRankingLoss(margin) (:470 of the same file builds it): the positive triples,
    repeated 2 * ratio times, against the corrupted copies behind them; L1 norm of head + relation - tail; target y = -1."""
    reps = int(valid_invalid_ratio_gat) * 2
    n_pos = int(train_indices.shape[0] / (reps + 1))                                        # :345-346
    pos = train_indices[:n_pos].repeat(reps, 1)                                             # :348-351
    neg = train_indices[n_pos:]
    pos_norm = torch.norm(entity_embed[pos[:, 0]] + relation_embed[pos[:, 1]] - entity_embed[pos[:, 2]], p=1, dim=1)      # :353-358
    neg_norm = torch.norm(entity_embed[neg[:, 0]] + relation_embed[neg[:, 1]] - entity_embed[neg[:, 2]], p=1, dim=1)      # :360-365
    y = -torch.ones(reps * n_pos, dtype=pos_norm.dtype)                                     # :367-370
    return torch.nn.functional.margin_ranking_loss(pos_norm, neg_norm, y, margin=margin)    # :372


def kg_graph(adj_indices, adj_values):
    """Corpus.get_graph (GAT/create_batch.py:708-732): graph[source][target] = [relations...] in insertion order;
    adj_indices [2,T] has the TARGET in row 0 and the SOURCE in row 1 (GAT/preprocess.py: rows = e2, cols = e1)."""
    graph = {}
    for t, s, r in zip(adj_indices[0].tolist(), adj_indices[1].tolist(), adj_values.tolist()):
        graph.setdefault(s, {}).setdefault(t, []).append(r)
    return graph


def kg_bfs(graph, source, nbd_size):
    """Corpus.bfs (:788-842): first-visit-wins breadth-first search; returns the nodes at EXACTLY `nbd_size` hops in
    discovery order as (relation lists along the path, target first ... source side last; entities target ... )."""
    visit, distance, parent = {source: 1}, {source: 0}, {source: (-1, -1)}
    queue = [source]
    while queue:
        top = queue.pop(0)
        for target in graph.get(top, {}):
            if target in visit:
                continue
            distance[target] = distance[top] + 1
            if distance[target] > nbd_size:
                continue                                                # :811-812 (not marked visited)
            queue.append(target)
            visit[target] = 1
            parent[target] = (top, graph[top][target])
    out = []
    for target in visit:
        if distance[target] != nbd_size:
            continue
        relations, entities, temp = [], [target], target
        while parent[temp] != (-1, -1):
            relations.append(parent[temp][1])
            entities.append(parent[temp][0])
            temp = parent[temp][0]
        out.append((tuple(tuple(r) for r in relations), tuple(entities[:-1])))
    return out


def kg_further_neighbors(graph, nbd_size):
    """Corpus.get_further_neighbors (:844-869): {source: [entries at nbd_size hops]} for every source that has out-edges."""
    res = {}
    for source in graph:
        n = kg_bfs(graph, source, nbd_size)
        if n:
            res[source] = n
    return res


def kg_batch_adj_data(neighbors_1hop, entities):
    """Corpus.get_batch_adj_data (:391-436): for every batch entity, in order, one edge (target, entity) per relation of every
    1-hop neighbour.  Returns (edge [2,E] int64 = (targets; sources), edge_type [E], source set, target set)."""
    trgts, srcs, vals, tset = [], [], [], set()
    for e in entities:
        for rel_tuple, target_tuple in neighbors_1hop.get(e, []):
            tset.add(target_tuple[0])
            for rel in rel_tuple[0]:
                trgts.append(target_tuple[0]); srcs.append(e); vals.append(rel)
    return (torch.tensor([trgts, srcs], dtype=torch.long).reshape(2, -1), torch.tensor(vals, dtype=torch.long), set(entities), tset)


def kg_batch_nhop_neighbors(neighbors_2hop, batch_sources, partial_2hop=False):
    """Corpus.get_batch_nhop_neighbors_all (:871-895): quadruples (source, first relation source->parent, first relation
    parent->target, target) for every 2-hop neighbour of every batch source, in the given source order."""
    quads = []
    for s in batch_sources:
        for i, (rels, ents) in enumerate(neighbors_2hop.get(s, [])):
            if partial_2hop and i >= 1:
                break
            quads.append([s, rels[-1][0], rels[0][0], ents[0]])
    return np.array(quads, dtype=np.int32).reshape(-1, 4)


# =============================================================================== GP-GNN side
def make_start_embedding(n, d):
    """utils/embedding_utils.py:170-182.  Channel c = ordered pair (i, j), i != j, row-major;
    ones in node i's first half-slot and node j's second half-slot.  Shape [n(n-1), 2dn, 1]."""
    C, S = n * (n - 1), 2 * d * n
    v = np.zeros((C, S, 1), dtype=np.float64)
    for c, (i, j) in enumerate((i, j) for i in range(n) for j in range(n) if i != j):
        v[c, 2 * d * i: 2 * d * i + d, 0] = 1.0
        v[c, 2 * d * j + d: 2 * d * (j + 1), 0] = 1.0
    return v


def get_head_indices(n, d, bs=50):
    """utils/embedding_utils.py:184-192 -> int64 [bs, C, 2d]: the 2d state slots of node i of pair (i, j)."""
    rows = [list(range(2 * d * i, 2 * d * (i + 1))) for i in range(n) for j in range(n) if i != j]
    return np.tile(np.asarray(rows, dtype=np.int64)[None], (bs, 1, 1))


def get_tail_indices(n, d, bs=50):
    """utils/embedding_utils.py:194-202 -> the 2d state slots of node j of pair (i, j)."""
    rows = [list(range(2 * d * j, 2 * d * (j + 1))) for i in range(n) for j in range(n) if i != j]
    return np.tile(np.asarray(rows, dtype=np.int64)[None], (bs, 1, 1))


def make_start_entity_embeddings(entity_embeddings, entity_pos_indices, embedding_dim, template, max_num_nodes=9):
    """utils/context_utils.py:387-426.  h0[b, c=(i,j)]: node i's first half-slot <- embedding of the
    pair's first entity, node j's second half-slot <- embedding of its second entity, zero elsewhere.
    (The reference pre-fills with the most frequent entity and masks with the template at the end;
    the result is this scatter.)"""
    n, d = max_num_nodes, embedding_dim
    B, C = entity_pos_indices.shape[:2]
    out = torch.zeros(B, C, 2 * d * n, 1, dtype=entity_embeddings.dtype)
    for c, (i, j) in enumerate((i, j) for i in range(n) for j in range(n) if i != j):
        out[:, c, 2 * d * i: 2 * d * i + d, 0] = entity_embeddings[entity_pos_indices[:, c, 0]]
        out[:, c, 2 * d * j + d: 2 * d * (j + 1), 0] = entity_embeddings[entity_pos_indices[:, c, 1]]
    return out * template


def build_block_adjacency(T, identity, n):
    """models/models.py:240-259 (copies :450-469, :660-679, :898-917).

    T: [B, n(n-1), (2d)^2] per-ordered-pair transition matrices (already through the non-linearity),
    identity: [2d, 2d].  Returns A [B, S, S] with A[b, i*2d+r, j*2d+c] = T[b, e(i,j)].view(2d,2d)[r, c]
    for i != j and identity[r, c] on the diagonal blocks.
    """
    B = T.shape[0]
    dd = identity.shape[0]
    blocks = torch.empty(B, n, n, dd, dd, dtype=T.dtype)
    e = 0
    for i in range(n):
        for j in range(n):
            if i == j:
                blocks[:, i, j] = identity
            else:
                blocks[:, i, j] = T[:, e].reshape(B, dd, dd)
                e += 1
    return blocks.permute(0, 1, 3, 2, 4).reshape(B, n * dd, n * dd)


def _stored(t, storage):
    """`t` as it reads back from a tensor of dtype `storage` (None: unchanged): the rounding a reference run on bfloat16 tensors applies
    to every tensor it materialises."""
    return t if storage is None else t.to(storage).to(t.dtype)


def propagate(adj_list, h0, nonlinearity, head_indices, tail_indices, as_gemm=False, storage=None, return_states=False):
    """models/models.py:260-274 (copies :470-485, :680-694, :918-932).

    storage (e.g. torch.bfloat16): every tensor the reference materialises per hop — the state after the non-linearity and the
    relation product — is rounded to that dtype, the arithmetic stays in the inputs' dtype (the reference's ops on bf16 tensors
    accumulate in fp32 and round the result once).  return_states: also the list of states [B, C, S] after each hop.

    adj_list: L tensors [B, S, S]; h0: [C, S, 1] (shared, GPGNN) or [B, C, S, 1] (per batch, RECON*);
    per hop h <- nonlinearity(A_l h); relation_l = gather(h, heads) * gather(h, tails);
    returns cat(relation_1..L, -1): [B, C, 2d*L].  head/tail_indices: [C, 2d] (or [B, C, 2d]).

    as_gemm: the same product written as ONE matrix product per graph (h^l[b] = A_l[b] . H[b], H = [S, C]) instead of the
    reference's broadcast of B*C matrix-vector products — identical maths (only the fp summation order inside the BLAS call may
    differ), used for sizes where the broadcast form takes minutes and tens of GB (n = 32: S = 512, C = 992);
    tests/test_oracle_golden.py checks it against the golden vectors like the default form.
    """
    B = adj_list[0].shape[0]
    h = h0
    rels, states = [], []
    hi = head_indices if head_indices.dim() == 3 else head_indices[None].expand(B, -1, -1)
    ti = tail_indices if tail_indices.dim() == 3 else tail_indices[None].expand(B, -1, -1)
    if as_gemm and h.dim() == 3:
        h = h[None].expand(B, -1, -1, -1)
    for A in adj_list:
        if as_gemm:
            h = torch.bmm(A, h.squeeze(-1).transpose(1, 2)).transpose(1, 2).unsqueeze(-1)      # [B, C, S, 1]
        else:
            h = torch.matmul(A[:, None], h)                 # [B, C, S, 1]
        if nonlinearity != "linear":
            h = getattr(F, nonlinearity)(h) if nonlinearity != "tanh" else torch.tanh(h)
        h = _stored(h, storage)
        flat = h.reshape(B, h.shape[1], h.shape[2])
        states.append(flat)
        rels.append(_stored(torch.gather(flat, 2, hi) * torch.gather(flat, 2, ti), storage))
    out = torch.cat(rels, dim=-1)
    return (out, states) if return_states else out


def propagate_backward(adj_list, h0, states, nonlinearity, head_indices, tail_indices, grad_out, storage=None):
    """models/models.py:260-274 differentiated by hand, from GIVEN states after each hop (so that a low-precision forward's own
    states — and with them its ReLU mask — can be used): per hop l = L..1
        Y_l      = (G_l + d relation_l / d h^l) . act'(h^l)        G_L = 0
        dA_l[b]  = Y_l[b]^T H^l-1[b]                                [S, S]
        G_l-1[b] = Y_l[b] A_l[b]                                    [C, S]
    with the relation term  R[c][head[c,x]] += g[c,x] h[c][tail[c,x]],  R[c][tail[c,x]] += g[c,x] h[c][head[c,x]].
    `storage` rounds Y_l, dA_l and G_l as a run on tensors of that dtype would.  Returns ([dA_1..dA_L], d loss / d h0 per graph [B, C, S]).
    tests/test_oracle_golden.py checks it against autograd of propagate()."""
    B, L = adj_list[0].shape[0], len(adj_list)
    dd = head_indices.shape[-1]
    hi = head_indices if head_indices.dim() == 3 else head_indices[None].expand(B, -1, -1)
    ti = tail_indices if tail_indices.dim() == 3 else tail_indices[None].expand(B, -1, -1)
    h0f = h0.reshape(B, h0.shape[1], h0.shape[2]) if h0.dim() == 4 else h0.reshape(1, h0.shape[0], h0.shape[1]).expand(B, -1, -1)      # as propagate() takes it
    G = torch.zeros_like(states[-1])
    g_adj = [None] * L
    for l in range(L, 0, -1):
        H = states[l - 1]
        g = grad_out[:, :, (l - 1) * dd: l * dd]
        R = torch.zeros_like(H)
        R.scatter_add_(2, hi, g * torch.gather(H, 2, ti))
        R.scatter_add_(2, ti, g * torch.gather(H, 2, hi))
        if nonlinearity == "relu":
            dact = (H > 0).to(H.dtype)
        elif nonlinearity == "tanh":
            dact = 1 - H * H
        else:
            dact = torch.ones_like(H)
        Y = _stored((G + R) * dact, storage)
        Hprev = states[l - 2] if l >= 2 else h0f
        g_adj[l - 1] = _stored(torch.bmm(Y.transpose(1, 2), Hprev), storage)
        G = _stored(torch.bmm(Y, adj_list[l - 1]), storage)
    return g_adj, G


def graph_convolution(x, adj, weight, bias=None):
    """GraphConvolution.forward   (models/layers.py:57-63): relu(adj @ (x @ W) + b).
    Accepts [n, in] / [n, n] (reference form) or batched [B, n, in] / [B, n, n]."""
    out = torch.matmul(adj, torch.matmul(x, weight))
    if bias is not None:
        out = out + bias
    return F.relu(out)


def build_adjecent_matrix(n, size=72):
    """utils/build_adjecent_matrix.py:6-17: line graph over ordered pairs of n entities, padded to
    72x72; A[x, y] = 1 if x[0]==y[1] or x[1]==y[0] or x==y; used rows divided by their sum."""
    A = np.zeros((size, size), dtype=np.float32)
    V = list(itertools.permutations(range(n), 2))
    for i, x in enumerate(V):
        for j, y in enumerate(V):
            A[i, j] = 1.0 if (x[0] == y[1] or x[1] == y[0] or i == j) else 0.0
        s = A[i].sum()
        if s != 0:
            A[i] /= s
    return A


# =============================================================================== synthetic workloads
def synthetic_batched_graph(B, n, e, F_, R, seed=0, n_rel=0):
    """SURVEY.md 8(d) synthetic generator: disjoint union of B graphs, n nodes / e edges each,
    dst/src uniform within a graph (duplicates and self loops allowed).  Returns CPU tensors."""
    g = torch.Generator().manual_seed(seed)
    base = (torch.arange(B) * n).repeat_interleave(e)
    dst = torch.randint(0, n, (B * e,), generator=g) + base
    src = torch.randint(0, n, (B * e,), generator=g) + base
    edge = torch.stack([dst, src])
    x = torch.randn(B * n, F_, generator=g)
    edge_embed = torch.randn(B * e, R, generator=g)
    return x, edge, edge_embed


def xavier_normal(shape, gain, generator):
    """nn.init.xavier_normal_ as used at GAT/layers.py:102-105 (std = gain*sqrt(2/(fan_in+fan_out)))."""
    fan_out, fan_in = shape
    std = gain * (2.0 / (fan_in + fan_out)) ** 0.5
    return torch.randn(shape, generator=generator) * std
