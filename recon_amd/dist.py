"""Data parallelism for the hot path: one process per GPU, whole graphs sharded
across ranks (no edge crosses graphs, so any partition of whole graphs is exact and needs no
data-path collective), parameter gradients summed with ONE all-reduce over a flat fp32 bucket
(torch.distributed backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests)."""
import os

import torch
import torch.distributed as dist

# RECON_DIST_FORCE=1: run every collective even at world size 1 (with a process group initialised) — the way to execute the RCCL calls of
# the data-parallel schedule (init with device_id, ReduceOp.AVG and its fallback, asynchronous handles under the backward, the bucket's
# per-region all-reduces) on a box with one GPU.  The results are unchanged (a mean over one rank).
FORCE_COLLECTIVES = os.environ.get("RECON_DIST_FORCE", "0") == "1"


def _collectives_on(group):
    return dist.is_available() and dist.is_initialized() and (dist.get_world_size(group) > 1 or FORCE_COLLECTIVES)


def shard_range(num_graphs, rank, world):
    """Contiguous block of graphs owned by `rank` (first `num_graphs % world` ranks get one extra)."""
    base, rem = divmod(num_graphs, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def shard_by_edges(edges_per_graph, world):
    """Greedy longest-processing-time bin packing of graphs by edge count (for power-law batches,
    SURVEY.md 8e).  Returns a list of graph-id lists, one per rank; deterministic."""
    order = sorted(range(len(edges_per_graph)), key=lambda i: (-int(edges_per_graph[i]), i))
    loads = [0] * world
    bins = [[] for _ in range(world)]
    for i in order:
        r = min(range(world), key=lambda k: (loads[k], k))
        bins[r].append(i)
        loads[r] += int(edges_per_graph[i])
    return [sorted(b) for b in bins]


def take_graph_shard(x, edge, edge_embed, node_ptr, edge_ptr, lo, hi):
    """Slice a batched disjoint-union graph to graphs [lo, hi): node rows and edge columns are
    contiguous slices; edge indices are rebased to the shard's first node."""
    n0, n1 = int(node_ptr[lo]), int(node_ptr[hi])
    e0, e1 = int(edge_ptr[lo]), int(edge_ptr[hi])
    return x[n0:n1], (edge[:, e0:e1] - n0), edge_embed[e0:e1]


class FlatGradBucket:
    """One flat fp32 buffer for all parameter gradients, reduced with a single all-reduce per step.

    zero() drops the gradients (so autograd ASSIGNS the freshly produced gradient tensors instead of
    launching one accumulate kernel per parameter); allreduce_mean() packs them into the flat buffer with
    one concatenation, averages across ranks, and leaves every p.grad as a view into the buffer."""

    def __init__(self, params, process_group=None):
        self.params = [p for p in params if p.requires_grad]
        self.group = process_group
        total = sum(p.numel() for p in self.params)
        dev = self.params[0].device if self.params else torch.device("cpu")
        self.flat = torch.zeros(total, dtype=torch.float32, device=dev)
        self.views = []
        self._avg_ok = True
        off = 0
        for p in self.params:
            self.views.append(self.flat[off:off + p.numel()].view_as(p))
            off += p.numel()

    def zero(self):
        for p in self.params:
            p.grad = None

    def region(self, params):
        """The piece of the flat buffer that holds `params` back to back in this order (a 1-d view), or None when they are not laid
        out that way.  A producer that writes its gradients THERE (models.SpGAT.write_head_gradients_into) saves pack() its copy:
        the gradients autograd assigns are already views of the buffer."""
        ids = {id(p): i for i, p in enumerate(self.params)}
        idx = [ids.get(id(p)) for p in params]
        if not idx or any(i is None for i in idx) or any(b != a + 1 for a, b in zip(idx, idx[1:])):
            return None
        off = sum(p.numel() for p in self.params[:idx[0]])
        return self.flat[off:off + sum(p.numel() for p in params)]

    def pack(self):
        """Bring every gradient into the flat buffer.  A gradient that already IS its view of the buffer (a second step
        without zero(), optimizer.zero_grad(set_to_none=False)) stays where it is; a missing one (unused parameter, empty
        shard) becomes zeros; the rest are copied with one multi-tensor copy.  Never writes through an aliased source, so
        no rank can fail here while the others wait in the collective."""
        dst, src = [], []
        for p, v in zip(self.params, self.views):
            g = p.grad
            if g is None:
                v.zero_()
            elif g.data_ptr() == v.data_ptr() and g.stride() == v.stride() and g.shape == v.shape:
                continue
            else:
                if g.dtype != v.dtype or g.device != v.device or g.shape != v.shape:
                    g = g.to(device=v.device, dtype=v.dtype).view_as(v)
                dst.append(v)
                src.append(g)
        if dst:
            torch._foreach_copy_(dst, src)
        for p, v in zip(self.params, self.views):
            p.grad = v

    def allreduce_mean(self, async_op=False, skip=None):
        """Average the bucket over the ranks.  `skip`: gradient tensors that ARE averaged already (OverlappedWeightGradSync.reduced: the
        attention layers' a / a_2, reduced inside their backward) — they are packed like the rest but left out of the collective, which
        then runs over the remaining contiguous pieces of the flat buffer (one all-reduce per piece; the same pieces on every rank, since
        the parameters and the layers that reduce in-backward are the same).  Everything else in the bucket (W, W_entities, embeddings ...)
        is averaged here: a bucket over a whole model needs this call, pack() alone averages nothing."""
        if not _collectives_on(self.group):
            self.pack()
            return None
        w = dist.get_world_size(self.group)
        done = set()
        if skip:
            # a parameter's gradient counts as averaged when it IS one of the marked tensors or lies inside one: the heads' backward marks
            # the fused [H, D, W] / [H, D] gradients, autograd hands each head its unbind() / split() view of them
            ids = {id(t) for t in skip}
            spans = [(t.untyped_storage().data_ptr(), t.data_ptr(), t.data_ptr() + t.numel() * t.element_size()) for t in skip if t.is_contiguous()]
            for i, p in enumerate(self.params):
                g = p.grad
                if g is None:
                    continue
                if id(g) in ids:
                    done.add(i)
                    continue
                if not g.is_contiguous():
                    continue
                st, lo, hi = g.untyped_storage().data_ptr(), g.data_ptr(), g.data_ptr() + g.numel() * g.element_size()
                if any(st == s0 and lo >= a0 and hi <= a1 for s0, a0, a1 in spans):
                    done.add(i)
        self.pack()
        if not done:
            return self._allreduce(self.flat, w, async_op)
        handles, off, start = [], 0, None
        for i, p in enumerate(self.params):                             # maximal runs of parameters still to be averaged
            if i in done:
                if start is not None:
                    handles.append(self._allreduce(self.flat[start:off], w, async_op))
                    start = None
            elif start is None:
                start = off
            off += p.numel()
        if start is not None:
            handles.append(self._allreduce(self.flat[start:off], w, async_op))
        return handles if async_op else None

    def _allreduce(self, t, w, async_op):
        if dist.get_backend(self.group) == "nccl" and self._avg_ok:    # RCCL averages inside the collective: no extra pass
            try:
                return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            except (RuntimeError, ValueError):                          # a build without ncclAvg: same on every rank
                self._avg_ok = False
        t.div_(w)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)


class OverlappedWeightGradSync:
    """Cross-rank averaging of the attention heads' weight gradients UNDER the backward's edge chain ("launch it as soon as
    the last layer's grads are ready, overlap with the remaining backward").

    g_a = G + a_2 (x) g_u with G = V^T g_h (the MFMA-bound product, 3.85 MB at cfg 2) and g_u = the score path's gradient (H x (2F+R)
    floats, known only after the edge chain).  Both enter linearly and a_2 is the same on every rank, so
        mean_ranks(g_a) = mean(G) + a_2 (x) mean(g_u),      mean_ranks(g_a_2) = a . mean(g_u):
    the backward (gat_layers._GATHeadsATPFunction, RECON_ATP_BWD_EARLY_SUM) sends G off asynchronously as soon as its split-K sum is
    done, runs the edge chain while it travels, reduces the few KB of g_u, and finishes on the means.  Gradients come back from autograd
    already averaged; `reduced` lists them so that a FlatGradBucket packs them without a second collective.

        sync = OverlappedWeightGradSync()
        with sync.installed():
            out.backward(G)
        bucket.allreduce_mean(skip=sync.reduced)     # averages what the backward did not (W, W_entities, embeddings ...); with a bucket of
                                                     # attention-layer parameters only, bucket.pack() is enough

    Every rank must take the same formulation of the layer: with more than one rank gat_layers.gat_path_for() chooses by the layer
    widths alone (never by this rank's E / N), so the collectives inside the backward match across ranks.
    world size 1 / no process group: every call is a no-op and the backward runs its usual single pass."""

    def __init__(self, process_group=None, force_sync=False):
        self.group = process_group
        self.force_sync = force_sync          # tests: the same arithmetic with blocking collectives (the serial schedule)
        self.reduced = []
        self._avg_ok = True

    def active(self):
        return _collectives_on(self.group)

    def all_reduce_mean(self, t, async_op=False):
        if not self.active():
            return None
        w = dist.get_world_size(self.group)
        async_op = async_op and not self.force_sync
        if dist.get_backend(self.group) == "nccl" and self._avg_ok:
            try:
                return dist.all_reduce(t, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            except (RuntimeError, ValueError):
                self._avg_ok = False
        t.div_(w)
        return dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def mark_reduced(self, *tensors):
        self.reduced.extend(t for t in tensors if t is not None)

    def installed(self):
        """Context manager: routes the heads' backward through this reducer while active (a no-op at world size 1)."""
        return _Installed(self)


class _Installed:
    def __init__(self, sync):
        self.sync, self.prev = sync, None

    def __enter__(self):
        from . import gat_layers
        self.sync.reduced = []
        self.prev = gat_layers.set_weight_grad_sync(self.sync if self.sync.active() else None)
        return self.sync

    def __exit__(self, *exc):
        from . import gat_layers
        gat_layers.set_weight_grad_sync(self.prev)
        return False


def overlapped_weight_grad_schedule(run_phase, g_big, g_small, sync):
    """The schedule of the heads' backward under an OverlappedWeightGradSync, with the compute phases as callables — what
    gat_layers._GATHeadsATPFunction.backward does with recon_gat_atp_bwd_phase; kept here in this form so that the CPU tests
    (gloo, world size 2) exercise the collectives' ordering and the linearity argument without a GPU."""
    run_phase("prepare")
    run_phase("weights_sum")                       # g_big <- G
    handle = sync.all_reduce_mean(g_big, async_op=True)
    run_phase("inputs")                            # g_small <- g_u   (must not touch g_big)
    sync.all_reduce_mean(g_small, async_op=False)
    if handle is not None:
        handle.wait()
    run_phase("finish")                            # g_big <- g_big + a_2 (x) g_small ; g_a_2 <- a . g_small
