#!/usr/bin/env python3
"""Stage-A batch assembly: the device-resident sampler against the reference's host algorithm (the oracle's
restatement of Corpus.get_batch_adj_data / get_batch_nhop_neighbors_all, dict walks over precomputed BFS neighbourhoods) on a
synthetic knowledge graph of FB15k-237's size (14 541 entities, 237 relations, 272 115 triples), 128 entities per batch."""
import json, os, sys, time
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))   # lives under tests/: it times the oracle as the host baseline
from recon_amd.sampler import KGNeighbourSampler
from oracle import recon_oracle as O           # the checker, timed here as the host baseline only

Ne, Tn, n_rel, B = 14541, 272115, 237, 128
rs = np.random.RandomState(0)
deg_w = 1.0 / np.arange(1, Ne + 1) ** 0.8                                  # skewed head distribution, as in a real KG
heads = rs.choice(Ne, size=Tn, p=deg_w / deg_w.sum())
tails = rs.randint(0, Ne, Tn)
adj = torch.from_numpy(np.stack([tails, heads])).long()
val = torch.from_numpy(rs.randint(0, n_rel, Tn)).long()
d = torch.device("cuda:0")
t0 = time.perf_counter(); sm = KGNeighbourSampler(adj.to(d), val.to(d), Ne); torch.cuda.synchronize(); t_build = time.perf_counter() - t0
ents = [rs.permutation(Ne)[:B].tolist() for _ in range(8)]
ents_d = [torch.tensor(e, device=d) for e in ents]
for e in ents_d[:2]:
    sm.batch_adj_data(e); sm.batch_nhop_neighbors(e)
torch.cuda.synchronize(); t0 = time.perf_counter()
sizes = []
for e in ents_d:
    (edge, et), _ = sm.batch_adj_data(e); q = sm.batch_nhop_neighbors(e); sizes.append((edge.shape[1], q.shape[0]))
torch.cuda.synchronize(); t_gpu = (time.perf_counter() - t0) / len(ents_d)
# host baseline: the graph dict and the two BFS passes are one-off preprocessing in the reference too; timed separately
t0 = time.perf_counter(); graph = O.kg_graph(adj, val); t_graph = time.perf_counter() - t0
sub = sorted({x for e in ents[:2] for x in e})
t0 = time.perf_counter(); n1 = {s: O.kg_bfs(graph, s, 1) for s in sub}; n2 = {s: O.kg_bfs(graph, s, 2) for s in sub}; t_bfs = (time.perf_counter() - t0) / len(sub)
t0 = time.perf_counter()
for e in ents[:2]:
    O.kg_batch_adj_data(n1, e); O.kg_batch_nhop_neighbors(n2, e)
t_host = (time.perf_counter() - t0) / 2
print(json.dumps({"workload": "stage-A batch assembly, FB15k-237-sized synthetic KG, 128 entities per batch",
                  "edges_per_batch": int(np.mean([s[0] for s in sizes])), "nhop_quads_per_batch": int(np.mean([s[1] for s in sizes])),
                  "gpu_ms_per_batch": 1e3 * t_gpu, "gpu_build_ms": 1e3 * t_build,
                  "host_ms_per_batch_dict_walks_only": 1e3 * t_host, "host_bfs_ms_per_entity_one_off": 1e3 * t_bfs,
                  "host_bfs_s_whole_graph_extrapolated": t_bfs * len(graph), "host_graph_dict_s": t_graph}))
