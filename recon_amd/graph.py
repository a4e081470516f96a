"""Graph preparation for the GAT hot path: the reference's int64 [2,E] COO edge tensors ->
destination-CSR / source-CSC index arrays on the device (K3, csrc/csr.hip).

The reference re-concatenates and re-coalesces the same edges in every one of the H+1 layer calls of
one SpGAT.forward (GAT/layers.py:124-127, :56-58); here the prepared graph is cached per
(edge tensor, n-hop tensor, N) so it is built once per batch.
"""
import ctypes as C
import os
from collections import OrderedDict

import torch

from . import _lib

_CACHE = OrderedDict()
_CACHE_MAX = 8
_VALIDATE = os.environ.get("RECON_VALIDATE_EDGES", "1") != "0"


def trust(*tensors, bound=None, rel_bound=None):
    """Mark index tensors whose values are in range BY CONSTRUCTION (produced on the device from already validated data: the
    neighbour sampler's batches, keys derived inside this package).  Validation costs a host synchronisation per fresh tensor
    (`aminmax` -> int()), and in the stage-A loop every iteration brings fresh tensors (GAT/main.py:478-516).  The mark records the
    tensor's version and the bound its values were validated against (`bound`: all values < bound; `rel_bound`: the second kind of id
    in a mixed tensor — the 2-hop quadruples hold entity and relation ids — handed down to the tensors derived from it): it is honoured
    only while the tensor has not been written to since, and only by a consumer whose own limit is not smaller.  Returns its argument(s)."""
    for t in tensors:
        if torch.is_tensor(t):
            t._recon_trusted = (t._version, None if bound is None else int(bound), None if rel_bound is None else int(rel_bound))
    return tensors[0] if len(tensors) == 1 else tensors


def trusted(t, limit=None):
    """Whether `t` carries a valid mark for a consumer that indexes `limit` rows: unmodified since it was marked, and validated against a
    bound <= limit (a mark without a bound — keys built inside this package for exactly the table they index — passes)."""
    m = getattr(t, "_recon_trusted", None)
    if m is None or m[0] != t._version:
        return False
    return limit is None or m[1] is None or m[1] <= int(limit)


def trust_bounds(t):
    m = getattr(t, "_recon_trusted", None)
    return (None, None) if m is None else (m[1], m[2])


_BAD = {}


_BAD_SLOTS = 4096


def _bad_flag(dev):
    """A zeroed int32 slot OF ITS OWN for every build (a ring of _BAD_SLOTS per device): the build kernels set it when they meet an id outside
    [0, N) and GraphCSR.resolve() reads it back with the hub counts.  A bad graph that is dropped unresolved leaves its slot set; the build
    that takes the slot a ring later pays one verification for it and clears it — nobody else ever sees it."""
    ring = _BAD.get(dev)
    if ring is None:
        ring = _BAD[dev] = [torch.zeros(_BAD_SLOTS, dtype=torch.int32, device=dev), 0]
    i = ring[1]
    ring[1] = (i + 1) % _BAD_SLOTS
    return ring[0][i:i + 1]


DEFER_HUB_READ = True   # the host read of a fresh graph's hub-table sizes (and of its id-range flag) waits until something needs them: see GraphCSR.resolve
HUB_CHUNK = 64          # RECON_HUB_CHUNK (include/recon_hip.h); 0 switches the splitting of long destination rows off (tests compare both)
# Row compaction (recon_graph.n_rows): a graph whose destination rows WITH edges are at most this fraction of its nodes — a knowledge-graph
# batch aggregates into the batch's ~128 entities of a 14 541-entity table (GAT/main.py:478-516) — gets the list of those rows, and the
# aggregate-then-project layer runs its node-parallel stages over them.  0 switches it off (tests compare both).
ROWS_COMPACT_MAX = float(os.environ.get("RECON_ROWS_COMPACT", "0.7"))


class GraphCSR:
    """Device-resident CSR/CSC view of a COO edge list.  Attributes mirror `recon_graph`."""

    def __init__(self, edge, N, rows_only=False, validate=True):
        """rows_only: destination CSR only — what the row sums need (SpecialSpmmFinal, the backward of gather_rows, the table gradients);
        no source view, no hub tables: half the sorting of a full build."""
        if not edge.is_cuda:
            raise RuntimeError("recon_amd: edge tensors must live on the GPU (no CPU path)")
        if edge.dtype != torch.int64 or edge.dim() != 2 or edge.shape[0] != 2:
            raise ValueError("edge must be an int64 [2,E] tensor")
        if edge.stride(1) != 1 and edge.shape[1] > 1:                     # rows must be dense; the two rows may lie apart (a column range of a wider [2, .] buffer)
            edge = edge.contiguous()
        E = edge.shape[1]
        if N >= 2 ** 31 or E >= 2 ** 31:
            raise ValueError("graph too large for int32 indices")
        dev = edge.device
        check = E > 0 and _VALIDATE and validate and not trusted(edge, N)
        # ids outside [0, N) would be truncated to int32, sorted on too few bits and make the edge kernels read out of bounds; the reference
        # fails on the same input (index out of range).  Full graphs with more than HUB_CHUNK edges are checked BY THE BUILD (a flag that comes
        # back with the hub counts: no round trip of its own); everything else with one aminmax here (one host sync per cached graph).
        in_build = check and not rows_only and HUB_CHUNK > 0 and E > HUB_CHUNK
        if check and not in_build:
            lo, hi = torch.aminmax(edge)
            lo, hi = int(lo), int(hi)
            if lo < 0 or hi >= N:
                raise IndexError("recon_amd: edge index out of range: ids span [%d, %d] but input has %d rows" % (lo, hi, N))
        self.N, self.E, self.device = int(N), int(E), dev
        self.edge = edge
        i32 = dict(dtype=torch.int32, device=dev)
        self.rows_only = bool(rows_only)
        L = _lib.lib()
        ws_bytes = L.recon_graph_workspace_bytes(self.N, self.E)
        # ONE allocation for the six index arrays and the sort workspace (a fresh graph per iteration — the reference's stage-A regime — is
        # host bound: seven allocations and two row views were a fifth of a build's 125 us); the arrays are handed out as views on first use
        a64 = lambda v: (v + 63) // 64 * 64                                  # 256-byte aligned parts, in int32 elements
        nr, ne = a64(N + 1), a64(max(E, 1))
        parts = [("rowptr_dst", N + 1, nr), ("eid", E, ne), ("dst", E, ne)]
        if not rows_only:
            parts += [("rowptr_src", N + 1, nr), ("src", E, ne), ("slot_by_src", E, ne)]
        total = sum(p_[2] for p_ in parts)
        self._buf = torch.empty(total + a64((ws_bytes + 3) // 4), **i32)
        self._parts, off = {}, 0
        for name, n_, room in parts:
            self._parts[name] = (off, n_)
            off += room
        base = self._buf.data_ptr()
        ptr = lambda name: base + 4 * self._parts[name][0] if name in self._parts else None
        self._c = _lib.ReconGraph(self.N, self.E, ptr("rowptr_dst"), ptr("eid"), ptr("src"), ptr("dst"), ptr("rowptr_src"), ptr("slot_by_src"))
        ws_ptr = base + 4 * total
        bad = _bad_flag(dev) if in_build else None
        hubs = HUB_CHUNK > 0 and E > HUB_CHUNK and not rows_only            # the hub-table sizes are counted by the build's last launch
        with _lib.on_device(dev):
            stream = _lib.current_stream()                                   # of `dev`, which need not be the current device
            if hubs:
                rc = L.recon_graph_build_counted(edge.data_ptr(), edge.data_ptr() + 8 * edge.stride(0), C.byref(self._c), ws_ptr, ws_bytes, _lib.ptr(bad), HUB_CHUNK, stream)
            else:
                rc = L.recon_graph_build_checked(edge.data_ptr(), edge.data_ptr() + 8 * edge.stride(0), C.byref(self._c), ws_ptr, ws_bytes, _lib.ptr(bad), stream)
        _lib.check(rc, "recon_graph_build")
        self._eid_long = None
        self._slot_idx = {}
        # hub rows (include/recon_hip.h): destinations with more than HUB_CHUNK in-edges are cut into pieces, one wavefront each.  Their
        # number comes back from the device — a host synchronisation, which with a fresh graph per iteration (the reference's regime) drains
        # the whole queue once per step and leaves the device idle until the host has launched again.  So the read waits (resolve()) until
        # something needs the sizes: the attention layer's score stage does not, and runs first (gat_layers._GATHeadsATPFunction).
        self._error = None
        self._pending = (ws_ptr, bad, stream) if hubs else None
        if not hubs:
            self.n_hub = self.n_piece = self.n_hub_src = self.n_piece_src = self.n_rows = 0
        elif not DEFER_HUB_READ:
            self.resolve()

    pending = property(lambda self: self._pending is not None)
    build_stream = property(lambda self: self._pending[2] if self._pending is not None else None)

    def raw_struct(self):
        """The `recon_graph` without hub tables, valid for work ordered behind the build on ITS stream that reads only the index arrays
        (recon_gat_atp_scores) — everything else goes through `.c` / call_struct(), which resolve first."""
        return self._c

    @property
    def c(self):
        self.resolve()
        return self._c

    def __getattr__(self, name):
        # the hub tables and their sizes exist once the counts have been read
        if name in ("n_hub", "n_piece", "n_hub_src", "n_piece_src", "hub_node", "hub_ptr", "piece", "hub_node_src", "hub_ptr_src", "piece_src", "n_rows") \
                and self.__dict__.get("_pending") is not None:
            self.resolve()
            if name in self.__dict__:
                return self.__dict__[name]
        raise AttributeError(name)

    def resolve(self):
        """Read the hub-table sizes (and the build's id-range flag) back and fill the tables: one host synchronisation on the build's stream."""
        if self._error is not None:
            raise IndexError(self._error)
        if self._pending is None:
            return
        ws_ptr, bad, stream = self._pending
        L, dev = _lib.lib(), self.device
        i32 = dict(dtype=torch.int32, device=dev)
        cnt = (C.c_int32 * 4)()
        bad_host, live = C.c_int32(0), C.c_int32(0)
        with _lib.on_device(dev):
            _lib.check(L.recon_graph_counts_read(C.byref(self._c), ws_ptr, cnt, C.byref(live), _lib.ptr(bad), C.byref(bad_host) if bad is not None else None, stream),
                       "recon_graph_counts_read")
        self._pending = None
        self.n_hub = self.n_piece = self.n_hub_src = self.n_piece_src = self.n_rows = 0     # n_rows = 0: rows are nodes
        if bad_host.value:
            # the slot is this build's own unless a bad graph that took it a ring ago was dropped unresolved: verify, and hand it back clear
            # (on the build's stream, which hubs_read has just drained: ordered in front of whichever build takes the slot next)
            with torch.cuda.stream(torch.cuda.ExternalStream(stream, device=dev) if stream else torch.cuda.default_stream(dev)):
                bad.zero_()
            lo, hi = (int(v) for v in torch.aminmax(self.edge))
            if lo < 0 or hi >= self.N:
                self._error = "recon_amd: edge index out of range: ids span [%d, %d] but input has %d rows" % (lo, hi, self.N)
                raise IndexError(self._error)
        if 0 < live.value <= ROWS_COMPACT_MAX * self.N:
            # few rows have edges: their list, their row pointers and the node -> row map (one launch on the build's stream, in front of the
            # hub tables, which then name rows)
            self.n_rows = int(live.value)
            self._rows = torch.empty(2 * self.n_rows + 1 + self.N, **i32)
            base = self._rows.data_ptr()
            self._c.n_rows, self._c.row_node, self._c.rowptr_rows = self.n_rows, base, base + 4 * self.n_rows
            self._c.node_row = base + 4 * (2 * self.n_rows + 1)
            with _lib.on_device(dev):
                _lib.check(L.recon_graph_rows_compact(C.byref(self._c), stream), "recon_graph_rows_compact")
                if not (cnt[0] > 0 or cnt[2] > 0) and _lib.current_stream() != stream:
                    torch.cuda.synchronize(dev)
        if cnt[0] > 0 or cnt[2] > 0:
            self.n_hub, self.n_piece, self.n_hub_src, self.n_piece_src = (int(v) for v in cnt)
            self._c.hub_chunk = HUB_CHUNK
            if self.n_hub:
                self.hub_node = torch.empty(self.n_hub, **i32)
                self.hub_ptr = torch.empty(self.n_hub + 1, **i32)
                self.piece = torch.empty(self.n_piece, 4, **i32)
                self._c.n_hub, self._c.n_piece = self.n_hub, self.n_piece
                self._c.hub_node, self._c.hub_ptr, self._c.piece = self.hub_node.data_ptr(), self.hub_ptr.data_ptr(), self.piece.data_ptr()
            if self.n_hub_src:
                self.hub_node_src = torch.empty(self.n_hub_src, **i32)
                self.hub_ptr_src = torch.empty(self.n_hub_src + 1, **i32)
                self.piece_src = torch.empty(self.n_piece_src, 4, **i32)
                self._c.n_hub_src, self._c.n_piece_src = self.n_hub_src, self.n_piece_src
                self._c.hub_node_src, self._c.hub_ptr_src, self._c.piece_src = (self.hub_node_src.data_ptr(), self.hub_ptr_src.data_ptr(),
                                                                                self.piece_src.data_ptr())
            with _lib.on_device(dev):
                _lib.check(L.recon_graph_hubs_fill(C.byref(self._c), stream), "recon_graph_hubs_fill")
                if _lib.current_stream() != stream:                       # resolved from another stream than the build's: its work must see the tables
                    torch.cuda.synchronize(dev)

    def _part(self, name):
        """View of one index array inside the graph's single allocation (None where a destination-only graph has none)."""
        hit = self._parts.get(name)
        return None if hit is None else self._buf[hit[0]:hit[0] + hit[1]]

    rowptr_dst = property(lambda self: self._part("rowptr_dst"))
    rowptr_src = property(lambda self: self._part("rowptr_src"))
    eid = property(lambda self: self._part("eid"))
    src = property(lambda self: self._part("src"))
    dst = property(lambda self: self._part("dst"))
    slot_by_src = property(lambda self: self._part("slot_by_src"))

    def slot_order_index(self, index, n_rows):
        """int32 [E]: index (int64 [E], original edge order) permuted to CSR-slot order — the table row each slot reads
        (recon_gat_atp_args.ee_index).  Cached per index tensor (identity + version): the two layers of a SpGAT share it."""
        key = (index.data_ptr(), index._version, tuple(index.stride()))
        hit = self._slot_idx.get(key)
        if hit is None:
            if _VALIDATE and self.E > 0 and not trusted(index, n_rows):
                lo, hi = torch.aminmax(index)
                if int(lo) < 0 or int(hi) >= n_rows:
                    raise IndexError("recon_amd: row index out of range: [%d, %d] into a table of %d rows" % (int(lo), int(hi), n_rows))
            idx = index.contiguous()
            slot32 = torch.empty(self.E, dtype=torch.int32, device=self.device)
            slot_long = torch.empty(self.E, dtype=torch.int64, device=self.device)
            with _lib.on_device(self.device):                                # index[eid] in both widths, one launch (was an index, a cast and eid's own cast)
                _lib.check(_lib.lib().recon_slot_index(idx.data_ptr(), self.eid.data_ptr(), self.E, slot32.data_ptr(), slot_long.data_ptr(), _lib.current_stream()),
                           "recon_slot_index")
            if trusted(index, n_rows):
                trust(slot_long, bound=trust_bounds(index)[0])                # a permutation of trusted values
            hit = (slot32, slot_long, index)                                 # `index` pins data_ptr identity while cached
            if len(self._slot_idx) >= 4:
                self._slot_idx.pop(next(iter(self._slot_idx)))
            self._slot_idx[key] = hit
        return hit[0]

    def slot_index_long(self, idx_slot):
        """The int64 twin of a slot_order_index() result (the segment key of the table-gradient row sum)."""
        for i32, i64, _ in self._slot_idx.values():
            if i32 is idx_slot:
                return i64
        return idx_slot.long()

    def call_struct(self, F, R, H):
        """The `recon_graph` to hand to ONE layer call of these sizes.  Graphs without hub rows: the cached struct itself (read-only).
        With hub rows the pieces' partial sums need scratch that the call writes: a fresh buffer per call (from the caching allocator,
        i.e. ordered on the calling stream) in a COPY of the struct — a cached graph stays read-only, so concurrent calls on one graph
        from different streams or threads do not share mutable state.  Returns (struct, keep-alive)."""
        if self.n_piece == 0 and self.n_piece_src == 0:
            return self.c, None
        need = _lib.lib().recon_graph_hub_ws_floats(C.byref(self.c), F, R, H)
        ws = torch.empty(need, dtype=torch.float32, device=self.device)
        c = _lib.ReconGraph.from_buffer_copy(self.c)
        c.hub_ws, c.hub_ws_floats = ws.data_ptr(), need
        return c, ws

    @property
    def eid_long(self):
        if self._eid_long is None:
            self._eid_long = self.eid.long()
        return self._eid_long


def _has_nhop(edge_list_nhop):
    # an absent n-hop arrives as a float tensor of shape [0] (GAT/models.py:57,81): test shape[0]
    return edge_list_nhop is not None and edge_list_nhop.shape[0] > 0


def prepare_graph(edge, edge_list_nhop, N, rows_only=False):
    """Concatenate 1-hop and n-hop edges (GAT/layers.py:124-127) and build / fetch the cached CSR."""
    nh = _has_nhop(edge_list_nhop)
    # identity + version + shape + strides: two views of one storage with equal shapes but different strides are different
    # edge lists.  (Writes through .data do not bump _version: do not mutate a cached edge tensor that way.)
    key = (edge.data_ptr(), edge._version, tuple(edge.shape), tuple(edge.stride()),
           edge_list_nhop.data_ptr() if nh else 0, edge_list_nhop._version if nh else 0,
           tuple(edge_list_nhop.shape) if nh else (), tuple(edge_list_nhop.stride()) if nh else (), int(N), str(edge.device), bool(rows_only),
           int(HUB_CHUNK))
    g = _CACHE.get(key)
    if g is not None:
        _CACHE.move_to_end(key)
        return g
    joined = getattr(edge, "_recon_joined", None)                       # (sampler.prune_batch: both lists are column ranges of one buffer, side by side)
    if nh and joined is not None and joined[1] is edge_list_nhop and joined[0].shape[1] == edge.shape[1] + edge_list_nhop.shape[1]:
        full = joined[0]
    else:
        full = torch.cat((edge, edge_list_nhop), dim=1) if nh else edge
    g = GraphCSR(full, N, rows_only, validate=not (trusted(edge, N) and (not nh or trusted(edge_list_nhop, N))))
    g._keepalive = (edge, edge_list_nhop if nh else None)   # pins data_ptr identity while cached
    _CACHE[key] = g
    while len(_CACHE) > _CACHE_MAX:
        _CACHE.popitem(last=False)
    return g


def clear_graph_cache():
    _CACHE.clear()
