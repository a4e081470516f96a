"""Synthetic inputs of the hot path: seeded, reproducible, CPU tensors.  Used by bench.py and
the tools/ scripts; the test infrastructure keeps its own copy of the same generators (tests assert they agree)."""
import torch


def synthetic_batched_graph(B, n, e, F_, R, seed=0):
    """Disjoint union of B graphs, n nodes / e edges each; dst/src uniform within a graph (duplicates and self
    loops allowed, the layer sums them).  Edge orientation as GAT/create_batch.py:429-433: row 0 = aggregation
    target, row 1 = neighbour.  Returns (x [B*n,F], edge int64 [2,B*e], edge_embed [B*e,R])."""
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
