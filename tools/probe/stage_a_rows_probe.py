"""How many destination rows of a stage-A batch graph have edges (the row-compaction decision of recon_amd.graph)?"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from stage_a_iter_bench import synthetic_kg
from recon_amd.sampler import KGNeighbourSampler
from recon_amd.graph import prepare_graph

dv = torch.device("cuda:0")
N = 14541
adj_idx, adj_val = synthetic_kg(N=N)
s = KGNeighbourSampler(adj_idx.to(dv), adj_val.to(dv), N)
src_all = torch.unique(adj_idx[1])
ents = src_all[torch.randperm(src_all.numel())[:128]].to(dv)
(edge, et), (srcs, _) = s.batch_adj_data(ents)
quads = s.batch_nhop_neighbors(srcs)
nh = torch.cat((quads[:, 3].unsqueeze(-1), quads[:, 0].unsqueeze(-1)), dim=1).t()
print("1-hop", edge.shape, "distinct dst", torch.unique(edge[0]).numel(), "distinct src", torch.unique(edge[1]).numel())
print("n-hop", nh.shape, "distinct dst", torch.unique(nh[0]).numel(), "distinct src", torch.unique(nh[1]).numel())
g = prepare_graph(edge, nh.contiguous(), N)
print("n_rows", g.n_rows, "n_hub", g.n_hub, "n_piece", g.n_piece, "n_hub_src", g.n_hub_src)
