"""On-disk formats either side of the hot path, so that stage A -> stage B hand-off works
with these layers without touching the reference's loaders:

  entity2id.txt / relation2id.txt     "<name> <id>" per line                       GAT/preprocess.py:6-28
  train.txt / valid.txt / test.txt    "<e1> <relation> <e2>" per line              GAT/preprocess.py:46-87
  entity2vec.txt / relation2vec.txt   one whitespace-separated float row per line  GAT/preprocess.py:30-43
  final_entity_embeddings.json, final_relation_embeddings.json
                                      {"<row index>": [floats]}, indent 4          GAT/main.py:406-413 (read: train.py / test.py json.load)
  W_ent2rel.json.npy                  numpy .npy (np.save appends .npy)            GAT_sep_space/main.py:982

Pure host code: no device work here."""
import json
import os

import numpy as np
import torch


def read_id_map(filename):
    """entity2id.txt / relation2id.txt -> {name: id}; lines with fewer than two fields are skipped."""
    out = {}
    with open(filename, "r") as f:
        for line in f:
            parts = line.strip().split()
            if len(parts) > 1:
                out[parts[0].strip()] = int(parts[1].strip())
    return out


read_entity_from_id = read_id_map          # GAT/preprocess.py:6
read_relation_from_id = read_id_map        # GAT/preprocess.py:18


def init_embeddings(entity_file, relation_file):
    """entity2vec.txt / relation2vec.txt -> two float32 arrays (GAT/preprocess.py:30-43)."""
    def rows(path):
        with open(path) as f:
            return np.array([[float(v) for v in line.strip().split()] for line in f], dtype=np.float32)
    return rows(entity_file), rows(relation_file)


def load_data(filename, entity2id, relation2id, is_unweigted=False, directed=True):
    """Triple file -> (triples [(e1, r, e2)], (rows, cols, data), unique entity names), GAT/preprocess.py:52-87:
    every triple contributes the adjacency entry (row = e2, col = e1, data = relation id or 1), and also the
    reverse entry when `directed` is False.  Blank lines are skipped."""
    triples, rows, cols, data = [], [], [], []
    unique = set()
    with open(filename) as f:
        for line in f:
            if line.strip() == "":
                continue
            e1, rel, e2 = (t.strip() for t in line.strip().split()[:3])
            unique.add(e1)
            unique.add(e2)
            triples.append((entity2id[e1], relation2id[rel], entity2id[e2]))
            w = 1 if is_unweigted else relation2id[rel]
            if not directed:
                rows.append(entity2id[e1]); cols.append(entity2id[e2]); data.append(w)
            rows.append(entity2id[e2]); cols.append(entity2id[e1]); data.append(w)
    return triples, (rows, cols, data), list(unique)


def edges_from_adjacency(adjacency):
    """(rows, cols, data) -> (edge_list int64 [2,E], edge_type int64 [E]) in the orientation the attention layer
    expects (GAT/create_batch.py:429-433 / SURVEY 8a G5): row 0 = aggregation target, row 1 = neighbour."""
    rows, cols, data = adjacency
    return torch.tensor([rows, cols], dtype=torch.int64), torch.tensor(data, dtype=torch.int64)


def save_embed(embeddings, save_path):
    """final_*_embeddings.json as GAT/main.py:406-413 writes it: {row index: [floats]}, indent 4."""
    arr = embeddings.detach().cpu().numpy() if isinstance(embeddings, torch.Tensor) else np.asarray(embeddings)
    with open(save_path, "w") as f:
        json.dump({idx: arr[idx].tolist() for idx in range(arr.shape[0])}, f, indent=4)


def load_embed(path):
    """Inverse of save_embed (the reference's consumers json.load the dict and index it by str(id)): float32 [rows, dim]."""
    with open(path) as f:
        d = json.load(f)
    n = len(d)
    return np.array([d[str(i)] for i in range(n)], dtype=np.float32)


def save_w_ent2rel(W, output_folder):
    """np.save(<folder>/W_ent2rel.json, W): numpy appends .npy (GAT_sep_space/main.py:982)."""
    arr = W.detach().cpu().numpy() if isinstance(W, torch.Tensor) else np.asarray(W)
    np.save(os.path.join(output_folder, "W_ent2rel.json"), arr)


def load_w_ent2rel(output_folder):
    return np.load(os.path.join(output_folder, "W_ent2rel.json.npy"))
