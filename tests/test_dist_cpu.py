"""Data-parallel path on CPU: 2 processes, gloo backend.  Whole graphs are sharded across ranks, each
rank computes the gradients of its shard (with the ORACLE standing in for the kernels — this test is
about sharding + the flat gradient bucket, not about the kernels), the bucket is all-reduced, and the
result must equal the single-process gradient of the full batch."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import recon_oracle as O
from recon_amd.dist import FlatGradBucket, shard_range, shard_by_edges, take_graph_shard


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _problem():
    B, n, e, F_, R, D = 6, 5, 12, 7, 6, 8
    x, edge, ee = O.synthetic_batched_graph(B, n, e, F_, R, seed=3)
    g = torch.Generator().manual_seed(0)
    a = O.xavier_normal((D, 2 * F_ + R), 1.414, g)
    a2 = O.xavier_normal((1, D), 1.414, g)
    G = torch.randn(B * n, D, generator=g)
    node_ptr = torch.arange(B + 1) * n
    edge_ptr = torch.arange(B + 1) * e
    return B, x, edge, ee, a, a2, G, node_ptr, edge_ptr


def _local_grads(x, edge, ee, a, a2, G, scale):
    a = torch.nn.Parameter(a.clone())
    a2 = torch.nn.Parameter(a2.clone())
    bucket = FlatGradBucket([a, a2])
    out = O.gat_layer_forward(x, edge, ee, None, None, a, a2, 0.2, True)
    ((out * G).sum() * scale).backward()
    return a, a2, bucket


def _worker(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    B, x, edge, ee, a, a2, G, node_ptr, edge_ptr = _problem()
    lo, hi = shard_range(B, rank, world)
    xs, es, ees = take_graph_shard(x, edge, ee, node_ptr, edge_ptr, lo, hi)
    Gs = G[int(node_ptr[lo]):int(node_ptr[hi])]
    # loss = sum over ALL graphs; allreduce_mean divides by world, so pre-scale by world
    pa, pa2, bucket = _local_grads(xs, es, ees, a, a2, Gs, float(world))
    bucket.allreduce_mean()
    assert pa.grad.data_ptr() == bucket.flat.data_ptr()          # after the reduction grads are views into the bucket
    ret[rank] = (pa.grad.clone().numpy(), pa2.grad.clone().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_gradient_equals_single_process():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    B, x, edge, ee, a, a2, G, _, _ = _problem()
    pa, pa2, _ = _local_grads(x, edge, ee, a, a2, G, 1.0)
    for r in range(world):
        np.testing.assert_allclose(ret[r][0], pa.grad.numpy(), rtol=1e-4, atol=1e-5)
        np.testing.assert_allclose(ret[r][1], pa2.grad.numpy(), rtol=1e-4, atol=1e-5)
    np.testing.assert_array_equal(ret[0][0], ret[1][0])           # every rank holds the same reduced bucket


def test_shard_helpers():
    assert [shard_range(10, r, 4) for r in range(4)] == [(0, 3), (3, 6), (6, 8), (8, 10)]
    assert shard_range(3, 3, 4) == (3, 3)                          # more ranks than graphs: empty shard
    bins = shard_by_edges([100, 1, 1, 1, 50, 50, 2], 2)
    assert sorted(sum(bins, [])) == list(range(7))
    loads = [sum([100, 1, 1, 1, 50, 50, 2][i] for i in b) for b in bins]
    assert abs(loads[0] - loads[1]) <= 5
    x = torch.arange(12.).view(6, 2)
    edge = torch.tensor([[0, 1, 2, 3, 4, 5], [1, 0, 3, 2, 5, 4]])
    ee = torch.arange(6.).view(6, 1)
    xs, es, ees = take_graph_shard(x, edge, ee, torch.tensor([0, 2, 4, 6]), torch.tensor([0, 2, 4, 6]), 1, 3)
    assert xs.shape == (4, 2) and es.tolist() == [[0, 1, 2, 3], [1, 0, 3, 2]] and ees.flatten().tolist() == [2., 3., 4., 5.]


def test_bucket_without_process_group_is_a_noop():
    p = torch.nn.Parameter(torch.ones(3))
    b = FlatGradBucket([p])
    (p * 2.0).sum().backward()
    assert b.allreduce_mean() is None and p.grad.tolist() == [2.0, 2.0, 2.0]
    b.pack()
    assert b.flat.tolist() == [2.0, 2.0, 2.0] and p.grad.data_ptr() == b.flat.data_ptr()
    b.zero()
    assert p.grad is None


def test_bucket_pack_none_grad_and_repeated_steps():
    """pack() must not write through an aliased source: an unused parameter (grad None) and a second step whose p.grad is
    already the bucket view (no zero() in between) both used to raise inside torch.cat(out=flat)."""
    used = torch.nn.Parameter(torch.ones(3))
    unused = torch.nn.Parameter(torch.ones(2, 2))
    b = FlatGradBucket([used, unused])
    (used * 2.0).sum().backward()
    b.pack()
    assert b.flat.tolist() == [2.0, 2.0, 2.0, 0.0, 0.0, 0.0, 0.0]
    assert unused.grad is not None and unused.grad.data_ptr() == b.views[1].data_ptr()
    # second step without zero(): autograd accumulates INTO the views, pack() must keep them
    (used * 3.0).sum().backward()
    b.pack()
    assert b.flat.tolist() == [5.0, 5.0, 5.0, 0.0, 0.0, 0.0, 0.0]
    # optimizer.zero_grad(set_to_none=False) keeps the views and zeroes them in place
    torch.optim.SGD([used, unused], lr=0.1).zero_grad(set_to_none=False)
    (used * 1.5).sum().backward()
    b.pack()
    assert b.flat.tolist() == [1.5, 1.5, 1.5, 0.0, 0.0, 0.0, 0.0]
    # a fresh (non-aliased) gradient after zero() is copied, stale content of the buffer is overwritten
    b.zero()
    (used * 4.0 + 0.0 * unused.sum()).sum().backward()
    b.pack()
    assert b.flat.tolist() == [4.0, 4.0, 4.0, 0.0, 0.0, 0.0, 0.0]


def _worker_empty_shard(rank, world, port, ret):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    p = torch.nn.Parameter(torch.arange(4.0))
    q = torch.nn.Parameter(torch.ones(2))
    bucket = FlatGradBucket([p, q])
    for step in range(2):                                      # two steps; rank 1 owns an empty shard (no backward at all)
        if rank == 0:
            ((p * p).sum() * float(world)).backward()           # q unused on every rank
        bucket.allreduce_mean()
        if step == 0:
            first = p.grad.clone()
            bucket.zero()
    ret[rank] = (first.numpy(), p.grad.clone().numpy(), q.grad.clone().numpy())
    dist.barrier()
    dist.destroy_process_group()


def test_two_rank_empty_shard_does_not_hang():
    world = 2
    mgr = mp.Manager()
    ret = mgr.dict()
    mp.spawn(_worker_empty_shard, args=(world, _free_port(), ret), nprocs=world, join=True)
    want = 2.0 * np.arange(4.0, dtype=np.float32)
    for r in range(world):
        np.testing.assert_allclose(ret[r][0], want)
        np.testing.assert_allclose(ret[r][1], want)
        np.testing.assert_array_equal(ret[r][2], np.zeros(2, np.float32))
