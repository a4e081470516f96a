"""Data parallelism for the hot path: one process per GPU, whole graphs sharded
across ranks (no edge crosses graphs, so any partition of whole graphs is exact and needs no
data-path collective), parameter gradients summed with ONE all-reduce over a flat fp32 bucket
(torch.distributed backend "nccl" = RCCL over xGMI; "gloo" in the CPU tests)."""
import torch
import torch.distributed as dist


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

    def allreduce_mean(self, async_op=False):
        if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size(self.group) == 1:
            return None
        w = dist.get_world_size(self.group)
        self.pack()
        if dist.get_backend(self.group) == "nccl" and self._avg_ok:    # RCCL averages inside the collective: no extra pass
            try:
                return dist.all_reduce(self.flat, op=dist.ReduceOp.AVG, group=self.group, async_op=async_op)
            except (RuntimeError, ValueError):                          # a build without ncclAvg: same on every rank
                self._avg_ok = False
        self.flat.div_(w)
        return dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group, async_op=async_op)
