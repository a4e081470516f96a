"""Data-parallel path on CPU: 2 processes, gloo backend.  Whole graphs are sharded across ranks, each
rank computes the gradients of its shard (with the ORACLE standing in for the kernels — this test is
about sharding + the flat gradient bucket, not about the kernels), the bucket is all-reduced, and the
result must equal the single-process gradient of the full batch."""
import json
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


# ------------------------------------------------------------------------------- the weight gradient reduced UNDER the edge chain
def _overlap_worker(rank, world, port, ret, force_sync):
    """recon_amd.dist.overlapped_weight_grad_schedule on two ranks, the compute phases played by the oracle's closed-form backward
    (GAT/layers.py:111-178 differentiated: g_a = G + a_2 (x) g_u with G = (k w gU[dst])^T edge_h, g_u = g_sigma^T edge_h,
    g_a_2 = g_u a^T): G is all-reduced asynchronously while "inputs" runs, g_u after it, "finish" works on the means."""
    from recon_amd.dist import OverlappedWeightGradSync, overlapped_weight_grad_schedule
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    B, x, edge, ee, a, a2, G, node_ptr, edge_ptr = _problem()
    lo, hi = shard_range(B, rank, world)
    xs, es, ees = take_graph_shard(x, edge, ee, node_ptr, edge_ptr, lo, hi)
    Gs = G[int(node_ptr[lo]):int(node_ptr[hi])] * float(world)              # loss = sum over ALL graphs; the reducer averages
    r = O.gat_layer_backward(xs, es, ees, None, None, a, a2, 0.2, True, Gs)
    edge_h = torch.cat((xs[es[0]], xs[es[1]], ees), dim=1)
    g_big, g_small, g_a2 = torch.empty_like(a), torch.empty(1, a.shape[1]), torch.empty_like(a2)
    log = []

    def run_phase(name):
        log.append(name)
        if name == "weights_sum":
            g_big.copy_((r["gm"] - r["g_sigma"][:, None] * a2[0][None, :]).t() @ edge_h)
        elif name == "inputs":
            g_small.copy_((r["g_sigma"][None, :] @ edge_h))
        elif name == "finish":
            g_big.add_(a2.t() @ g_small)
            g_a2.copy_(g_small @ a.t())
    sync = OverlappedWeightGradSync(force_sync=force_sync)
    assert sync.active()
    overlapped_weight_grad_schedule(run_phase, g_big, g_small, sync)
    assert log == ["prepare", "weights_sum", "inputs", "finish"]
    ret[rank] = (g_big.clone().numpy(), g_a2.clone().numpy(), r["g_a"].numpy(), r["g_a_2"].numpy())
    dist.barrier()
    dist.destroy_process_group()


def _run_overlap(force_sync):
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_overlap_worker, args=(world, _free_port(), ret, force_sync), nprocs=world, join=True)
    return ret


def test_overlapped_weight_gradient_schedule_world2():
    """(i) the overlapped schedule (asynchronous all-reduce of G under "inputs") gives BIT-equal gradients to the same arithmetic with
    blocking collectives; (ii) both ranks end with the same values; (iii) they equal the mean over ranks of the full local
    gradients (the plain one-bucket all-reduce) and the single-process gradient of the whole batch."""
    over, serial = _run_overlap(False), _run_overlap(True)
    for k in range(2):
        np.testing.assert_array_equal(over[0][k], over[1][k])
        np.testing.assert_array_equal(over[0][k], serial[0][k])
        np.testing.assert_array_equal(over[1][k], serial[1][k])
    plain_a = (over[0][2] + over[1][2]) / 2                                 # mean of the ranks' complete g_a / g_a_2
    plain_a2 = (over[0][3] + over[1][3]) / 2
    np.testing.assert_allclose(over[0][0], plain_a, atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(over[0][1], plain_a2, atol=1e-5, rtol=1e-5)
    B, x, edge, ee, a, a2, G, _, _ = _problem()
    full = O.gat_layer_backward(x, edge, ee, None, None, a, a2, 0.2, True, G)
    np.testing.assert_allclose(over[0][0], full["g_a"].numpy(), atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(over[0][1], full["g_a_2"].numpy(), atol=1e-5, rtol=1e-5)


def test_weight_grad_sync_is_inert_without_a_process_group():
    from recon_amd.dist import OverlappedWeightGradSync
    from recon_amd import gat_layers
    sync = OverlappedWeightGradSync()
    assert not sync.active() and sync.all_reduce_mean(torch.ones(3)) is None
    with sync.installed():
        assert gat_layers._WEIGHT_GRAD_SYNC is None                        # world size 1: the backward keeps its single pass


# ------------------------------------------------------------------------------- the FULL stack's bucket (heads + out_att + a non-attention parameter)
def _full_stack_worker(rank, world, port, ret, mode):
    """SpGAT's parameter set in ONE FlatGradBucket: two attention layers (the heads, then out_att — backward order: out_att first) whose
    a / a_2 are averaged inside their backward (overlapped_weight_grad_schedule: the big term asynchronously under the edge chain), plus a
    parameter no attention backward reduces (W of GAT/models.py:75).  mode 'overlap': asynchronous collectives + allreduce_mean(skip=
    sync.reduced); 'serial': the same with blocking collectives; 'plain': nothing in-backward, one flat all-reduce of everything.
    The compute phases are played by the oracle's closed-form backward (the kernels are tested on the GPU)."""
    from recon_amd.dist import OverlappedWeightGradSync, overlapped_weight_grad_schedule
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    B, x, edge, ee, a, a2, G, node_ptr, edge_ptr = _problem()
    lo, hi = shard_range(B, rank, world)
    xs, es, ees = take_graph_shard(x, edge, ee, node_ptr, edge_ptr, lo, hi)
    Gs = G[int(node_ptr[lo]):int(node_ptr[hi])] * float(world)
    gen = torch.Generator().manual_seed(7)
    a_o, a2_o = O.xavier_normal(tuple(a.shape), 1.414, gen), O.xavier_normal(tuple(a2.shape), 1.414, gen)
    layers = []
    for (la, la2, scale) in ((a_o, a2_o, 0.5), (a, a2, 1.0)):              # backward order: out_att, then the heads
        r = O.gat_layer_backward(xs, es, ees, None, None, la, la2, 0.2, True, Gs * scale)
        layers.append((la, la2, r, torch.cat((xs[es[0]], xs[es[1]], ees), dim=1)))
    W = torch.nn.Parameter(torch.arange(6.0).view(2, 3))
    params = [torch.nn.Parameter(a.clone()), torch.nn.Parameter(a2.clone()), W, torch.nn.Parameter(a_o.clone()), torch.nn.Parameter(a2_o.clone())]
    bucket = FlatGradBucket(params)                                           # W sits BETWEEN the attention parameters: two pieces to reduce
    sync = OverlappedWeightGradSync(force_sync=(mode == "serial"))
    bucket.zero()
    sync.reduced = []
    W.grad = torch.full((2, 3), float(rank + 1)) * float(world)
    for li, (la, la2, r, edge_h) in enumerate(layers):
        g_big, g_small, g_a2 = torch.empty_like(la), torch.empty(1, la.shape[1]), torch.empty_like(la2)

        def run_phase(name):
            if name == "weights_sum":
                g_big.copy_((r["gm"] - r["g_sigma"][:, None] * la2[0][None, :]).t() @ edge_h)
            elif name == "inputs":
                g_small.copy_((r["g_sigma"][None, :] @ edge_h))
            elif name == "finish":
                g_big.add_(la2.t() @ g_small)
                g_a2.copy_(g_small @ la.t())
        if mode == "plain":
            for name in ("prepare", "weights_sum", "inputs", "finish"):
                run_phase(name)
        else:
            overlapped_weight_grad_schedule(run_phase, g_big, g_small, sync)
            sync.mark_reduced(g_big, g_a2)
        pa, pa2 = (params[3], params[4]) if li == 0 else (params[0], params[1])
        pa.grad, pa2.grad = g_big, g_a2
    bucket.allreduce_mean(skip=None if mode == "plain" else sync.reduced)
    assert all(p.grad.data_ptr() == v.data_ptr() for p, v in zip(bucket.params, bucket.views))
    ret[rank] = [p.grad.clone().numpy() for p in params]
    dist.barrier()
    dist.destroy_process_group()


def _run_full_stack(mode):
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_full_stack_worker, args=(world, _free_port(), ret, mode), nprocs=world, join=True)
    return ret


def test_full_stack_bucket_overlapped_equals_blocking_and_plain():
    """A bucket over a WHOLE model under the overlapped schedule: (i) bit-equal to the same schedule with blocking collectives, (ii) the
    same on both ranks, (iii) the parameter no attention backward touches (W) IS averaged (pack() alone would leave every rank its own
    value — the advisor's finding), (iv) equal to the plain flat all-reduce of the complete local gradients."""
    over, serial, plain = _run_full_stack("overlap"), _run_full_stack("serial"), _run_full_stack("plain")
    for k in range(5):
        np.testing.assert_array_equal(over[0][k], over[1][k])
        np.testing.assert_array_equal(over[0][k], serial[0][k])
        np.testing.assert_allclose(over[0][k], plain[0][k], atol=1e-5, rtol=1e-5)
    np.testing.assert_array_equal(over[0][2], np.full((2, 3), 3.0, np.float32))     # mean over ranks of world * (rank + 1) = 2 * 1.5


def _path_worker(rank, world, port, ret):
    from recon_amd import gat_layers
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    before = gat_layers._data_parallel()
    dist.init_process_group("gloo", rank=rank, world_size=world)
    ret[rank] = (before, gat_layers._data_parallel())
    dist.barrier()
    dist.destroy_process_group()


def test_formulation_choice_ignores_the_local_batch_under_data_parallelism(monkeypatch):
    """gat_path_for() must give every rank the same answer: with more than one rank it may not look at this rank's E / N (one rank on
    'atp' — two collectives inside its backward — and another on 'proj' — none — would hang).  The kernels' own support query is a
    function of the widths only (csrc/gat_atp.hip: recon_gat_atp_supported ignores N and E)."""
    from recon_amd import gat_layers

    class _Lib:
        @staticmethod
        def recon_gat_atp_supported(N, E, F, R, D, H):
            return 1
    monkeypatch.setattr(gat_layers._lib, "lib", lambda: _Lib)
    monkeypatch.setattr(gat_layers, "_GAT_PATH", "auto")
    assert gat_layers.gat_path_for(1000, 60, 8, 8, 8, 2) == "proj"           # one process: sparse batches project first ...
    assert gat_layers.gat_path_for(1000, 100, 8, 8, 8, 2) == "atp"           # ... unless the graph's rows with edges get compacted (more than HUB_CHUNK edges)
    from recon_amd import graph as graph_mod
    monkeypatch.setattr(graph_mod, "ROWS_COMPACT_MAX", 0.0)
    assert gat_layers.gat_path_for(1000, 100, 8, 8, 8, 2) == "proj"
    monkeypatch.setattr(gat_layers, "_data_parallel", lambda: True)
    assert gat_layers.gat_path_for(1000, 100, 8, 8, 8, 2) == "atp" and gat_layers.gat_path_for(10, 100, 8, 8, 8, 2) == "atp"
    monkeypatch.undo()
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_path_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    assert all(ret[r] == (False, True) for r in range(world))


def test_bucket_region_finds_consecutive_parameters():
    """FlatGradBucket.region: the flat piece holding a run of parameters in bucket order, None for anything else."""
    ps = [torch.nn.Parameter(torch.zeros(n)) for n in (3, 5, 2, 4)]
    b = FlatGradBucket(ps)
    r = b.region(ps[1:3])
    assert r is not None and r.data_ptr() == b.flat.data_ptr() + 4 * 3 and r.numel() == 7
    assert b.region(ps).numel() == 14 and b.region([ps[3]]).numel() == 4
    assert b.region([ps[0], ps[2]]) is None and b.region([ps[1], ps[0]]) is None and b.region([]) is None
    assert b.region([torch.nn.Parameter(torch.zeros(3))]) is None


def test_weight_grad_destination_is_handed_out_once_per_release():
    """gat_layers._weight_grad_tensors decides at BACKWARD time: the registered destination goes to the first backward after a release,
    every later one gets fresh tensors (two forwards before one backward; accumulation) — ADVICE r4."""
    from recon_amd import gat_layers as GL
    H, D, W = 2, 3, 5
    a = torch.zeros(H, D, W)
    ga, ga2 = torch.zeros(H, D, W), torch.zeros(H, D)
    f32 = dict(dtype=torch.float32, device=a.device)
    GL.set_weight_grad_destination(a, ga, ga2)                        # forward 1 (every .grad None): release
    GL.set_weight_grad_destination(a, ga, ga2)                        # forward 2 before any backward: release again
    t1 = GL._weight_grad_tensors(a, H, D, W, f32)
    t2 = GL._weight_grad_tensors(a, H, D, W, f32)
    assert t1[0] is ga and t1[1] is ga2
    assert t2[0] is not ga and t2[0].data_ptr() != ga.data_ptr() and t2[1].data_ptr() != ga2.data_ptr()
    GL.set_weight_grad_destination(a, ga, ga2, release=False)         # a forward that finds gradients in place: the claim stands
    assert GL._weight_grad_tensors(a, H, D, W, f32)[0] is not ga
    GL.set_weight_grad_destination(a, ga, ga2, release=True)          # gradients dropped: the region is free again
    assert GL._weight_grad_tensors(a, H, D, W, f32)[0] is ga
    GL.set_weight_grad_destination(a, None, None)
    assert GL._weight_grad_tensors(a, H, D, W, f32)[0] is not ga


def _skip_views_worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from recon_amd.dist import FlatGradBucket
        H, D, W = 3, 2, 4
        heads_a = [torch.nn.Parameter(torch.zeros(D, W)) for _ in range(H)]
        heads_a2 = [torch.nn.Parameter(torch.zeros(1, D)) for _ in range(H)]
        other = torch.nn.Parameter(torch.zeros(5))
        bucket = FlatGradBucket(heads_a + heads_a2 + [other])
        # what the heads' backward marks: the fused gradients (already averaged: the same on both ranks); what the parameters hold:
        # their unbind() / split() views
        g_a = torch.arange(H * D * W, dtype=torch.float32).view(H, D, W) + 1.0
        g_a2 = torch.arange(H * D, dtype=torch.float32).view(H, D) + 100.0
        for p, g in zip(heads_a, g_a.unbind(0)):
            p.grad = g
        for p, g in zip(heads_a2, g_a2.split(1, 0)):
            p.grad = g
        other.grad = torch.full((5,), float(rank + 1))
        calls = []
        orig = bucket._allreduce

        def counting(t, w, async_op):
            calls.append(t.numel())
            return orig(t, w, async_op)
        bucket._allreduce = counting
        bucket.allreduce_mean(skip=[g_a, g_a2])
        ok = (calls == [5] and torch.equal(torch.stack([p.grad for p in heads_a]), g_a) and
              torch.equal(torch.cat([p.grad for p in heads_a2]), g_a2) and torch.allclose(other.grad, torch.full((5,), 1.5)))
        q.put((rank, ok, calls))
    finally:
        dist.destroy_process_group()


def test_bucket_skip_matches_per_head_views_of_the_marked_gradients():
    """allreduce_mean(skip=sync.reduced): the heads' backward marks the fused [H, D, W] / [H, D] gradients, the parameters hold unbind() /
    split() views of them — they must be recognised (storage range), or the 3.85 MB term is all-reduced a second time."""
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    ps = [ctx.Process(target=_skip_views_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in ps:
        p.start()
    res = [q.get(timeout=120) for _ in ps]
    for p in ps:
        p.join(60)
    assert all(ok for _, ok, _ in res), res


def test_bench_launcher_relays_the_result_line_and_the_exit_code(tmp_path, capsys):
    """`python bench.py --gpus N` outside torch.distributed.run starts the ranks as a child process (SURVEY 8e / VERDICT r4): rank 0's JSON
    line comes back on stdout, a failing rank makes the launcher fail, and so does a run that ends without a result line."""
    import argparse
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    script = tmp_path / "fake_rank.py"
    script.write_text(
        "import os, sys, json\n"
        "mode = sys.argv[sys.argv.index('--mode') + 1]\n"
        "rank, world = int(os.environ['RANK']), int(os.environ['WORLD_SIZE'])\n"
        "assert world == 2 and os.environ['MASTER_ADDR'] == '127.0.0.1'\n"
        "print('noise from rank %d' % rank)\n"
        "if mode == 'fail' and rank == 1: sys.exit(3)\n"
        "if mode != 'silent' and rank == 0: print(json.dumps({'metric': 'm', 'value': 1.0, 'n_gpus': world}))\n")
    args = argparse.Namespace(gpus=2)
    rc = bench.launch_ranks(args, argv=["--mode", "ok"], script=str(script))
    out = capsys.readouterr()
    lines = [l for l in out.out.splitlines() if l.strip()]
    assert rc == 0 and len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2 and "noise from rank" in out.err
    assert bench.launch_ranks(args, argv=["--mode", "fail"], script=str(script)) != 0
    capsys.readouterr()
    assert bench.launch_ranks(args, argv=["--mode", "silent"], script=str(script)) == 1


def _scaling_fields_worker(rank, world, port, ret):
    import sys
    import time
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import bench
        t = torch.ones(1 << 16)
        calls = {"comm": 0, "local": 0}

        def step_no_comm():
            calls["local"] += 1
            time.sleep(0.002 * (rank + 1))                              # rank 1 is the straggler: the MAX over ranks must carry its time

        def step():
            calls["comm"] += 1
            time.sleep(0.002 * (rank + 1))
            dist.all_reduce(t)

        def reduce_max(v):
            x = torch.tensor([v], dtype=torch.float64)
            dist.all_reduce(x, op=dist.ReduceOp.MAX)
            return float(x.item())
        steps, E = 10, 1000
        dt0 = bench.timed_region(step_no_comm, steps, dist.barrier, reduce_max)
        dt1 = bench.timed_region(step, steps, dist.barrier, reduce_max)
        f = bench.scaling_fields(E, steps, dt1, dt0)
        ret[rank] = (dict(f), dt0, dt1, dict(calls))
    finally:
        dist.destroy_process_group()


def test_bench_line_carries_a_no_collective_baseline_at_world_size_2():
    """VERDICT r5 #2 / SURVEY 8e: at N > 1 the bench line holds the SAME steps timed with every collective skipped (same process, max over
    ranks), so that one run yields communication overhead and an efficiency that does not mix shard sizes."""
    world = 2
    ret = mp.get_context("spawn").Manager().dict()
    mp.spawn(_scaling_fields_worker, args=(world, _free_port(), ret), nprocs=world, join=True)
    f0, dt0, dt1, calls = ret[0]
    assert ret[1][1] == dt0 and ret[1][2] == dt1                        # max over ranks: every rank holds the same two times
    assert calls == {"comm": 10, "local": 10}                           # EXACTLY `steps` calls inside each timed region
    assert set(f0) == {"per_gpu_no_comm_edges_per_s", "per_gpu_no_comm_ms_per_step", "comm_overhead_ms", "efficiency_vs_no_comm"}
    assert dt0 >= 10 * 0.004 * 0.95                                     # the straggler's 4 ms per step, not rank 0's 2
    assert abs(f0["per_gpu_no_comm_edges_per_s"] - 1000 * 10 / dt0) < 1e-6 * f0["per_gpu_no_comm_edges_per_s"]
    assert abs(f0["comm_overhead_ms"] - 1e3 * (dt1 - dt0) / 10) < 1e-9 and abs(f0["efficiency_vs_no_comm"] - dt0 / dt1) < 1e-12
    assert 0.2 < f0["efficiency_vs_no_comm"] < 1.2
