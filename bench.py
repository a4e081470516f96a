#!/usr/bin/env python3
"""bench.py — edges aggregated / second, GAT forward + backward, on synthetic KG-context graphs
(BASELINE.json metric).  One step = one pass of the H-head attention stage (GAT/models.py:71-72
equivalent: H SpGraphAttentionLayers sharing inputs, outputs concatenated) forward + backward over
one batch of synthetic graphs, inputs resident in HBM.  Default workload = BASELINE.json configs[1]:
512 graphs x 16 nodes x 64 edges, F = R = 200, 8 heads x D = 200, fp32.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...

N > 1: one process per GPU, every rank owns its own graphs (weak scaling) — BASELINE.json configs[3]: 8 192 graphs sharded over 8 GPUs,
i.e. 1 024 graphs per GPU, is the default whenever --gpus > 1 (`--workload cfg4`; N = 1 runs configs[1], 512 graphs); `--workload cfg5`
runs configs[4]'s power-law graphs (up to 256 nodes / 4 096 edges each), dealt to the ranks by edge count (recon_amd.dist.shard_by_edges).
The parameter gradients are averaged over RCCL with the big term (G = V^T g_h, 3.85 MB) sent off before the backward's
edge chain and the few KB that depend on it afterwards (recon_amd/dist.py).  Rank 0 prints ONE JSON line.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

HBM_PEAK = 8.0e12          # B/s, MI355X spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_F32_PEAK = 157.3e12   # FLOP/s, fp32-input MFMA
MFMA_BF16_PEAK = 2.5e15    # FLOP/s, dense bf16 / f16 MFMA (the split-precision GEMMs issue 3 f16 or 6 bf16 MFMAs per fp32 product)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--graphs", type=int, default=None, help="graphs per GPU (default: 512 on one GPU = configs[1], 1024 per GPU otherwise = configs[3])")
    ap.add_argument("--workload", choices=("auto", "cfg2", "cfg4", "cfg5"), default="auto",
                    help="auto: cfg2 on one GPU, cfg4 (8192 graphs / 8 GPUs = 1024 per GPU) on several; cfg5: power-law graphs sharded by edge count")
    ap.add_argument("--nodes", type=int, default=16)
    ap.add_argument("--edges", type=int, default=64, help="edges per graph")
    ap.add_argument("--feat", type=int, default=200, help="F = R")
    ap.add_argument("--dim", type=int, default=200, help="D per head")
    ap.add_argument("--heads", type=int, default=8)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--no-extras", action="store_true", help="main step and roofline only (profiling runs): no exact_fp32 / uncached / secondary legs")
    return ap.parse_args()


def _free_port():
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def launch_ranks(args, argv=None, script=None):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start `python -m torch.distributed.run --nnodes=1 --nproc-per-node N
    bench.py <same arguments>` as a CHILD process (this process has not touched the GPU and never will: a process that has must not be
    replaced), relay rank 0's JSON line on stdout (everything else the ranks print goes to stderr) and return the child's exit code —
    non-zero too when the ranks ended without a result line."""
    import subprocess
    argv = sys.argv[1:] if argv is None else argv
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), script or os.path.abspath(__file__)] + list(argv)
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, text=True)
    line = None
    for out in proc.stdout:
        if out.startswith("{") and '"metric"' in out:
            line = out.strip()
        else:
            sys.stderr.write(out)
    rc = proc.wait()
    if line is not None:
        print(line, flush=True)
    return rc if rc != 0 else (0 if line is not None else 1)


def timed_region(step, steps, barrier_sync, reduce_max=None):
    """EXACTLY `steps` calls of `step` bracketed by barrier + device synchronisation on both sides; seconds, MAX over the ranks."""
    barrier_sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    barrier_sync()
    dt = time.perf_counter() - t0
    return reduce_max(dt) if reduce_max is not None else dt


def scaling_fields(edges_per_gpu, steps, dt_comm, dt_no_comm):
    """What an N > 1 line carries so that ONE run separates communication cost from shard-size effects: the same step with
    every collective skipped, timed in the same process right before the measured loop (max over ranks, as the measured loop is)."""
    return {"per_gpu_no_comm_edges_per_s": edges_per_gpu * steps / dt_no_comm,
            "per_gpu_no_comm_ms_per_step": 1e3 * dt_no_comm / steps,
            "comm_overhead_ms": 1e3 * (dt_comm - dt_no_comm) / steps,
            "efficiency_vs_no_comm": dt_no_comm / dt_comm}


def algorithmic_bytes_fwd(N, E, H, D):
    """bytes the fused forward edge-aggregation must move (fp32 values, int32 CSR):
    Q [E,HD] once, P_dst and P_src [N,HD] once each, out [N,HD] once, CSR, a_2."""
    HD = H * D
    return 4 * HD * (E + 3 * N) + 4 * (E + N + 1) + 4 * HD


def main():
    args = parse()
    if "WORLD_SIZE" not in os.environ and args.gpus > 1:
        sys.exit(launch_ranks(args))                         # before anything touches the GPU: the ranks are child processes
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        sys.exit("bench.py --gpus %d under WORLD_SIZE=%d: the two must agree" % (args.gpus, world))
    ndev = max(1, torch.cuda.device_count())                 # counting devices does not initialise the GPU
    force = os.environ.get("RECON_DIST_FORCE", "0") == "1"   # world size 1 with every collective executed (recon_amd/dist.py)
    backend = None
    if world > 1 or force:
        # "nccl" = RCCL over xGMI.  Fewer GPUs than ranks (a 1-GPU box): RCCL refuses two ranks on one device, so the ranks share the GPU
        # and talk over gloo — exercises the launcher and the schedule, measures nothing.  RECON_DIST_BACKEND overrides.
        backend = os.environ.get("RECON_DIST_BACKEND", "nccl" if ndev >= world else "gloo")
    torch.cuda.set_device(local_rank % ndev)                 # one GPU per rank on a real node (ndev >= world)
    dev = torch.device("cuda", local_rank % ndev)
    if backend is not None:
        if world == 1:
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            os.environ.setdefault("MASTER_PORT", str(_free_port()))
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=dev, rank=rank, world_size=world)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    # One process per GPU: the autograd engine's per-device worker threads buy nothing and the hand-over costs ~0.2 ms of host
    # time per backward call — as much as 40 % of this step's GPU time (measured: host enqueue 0.40 -> 0.21 ms/step).
    torch.autograd.set_multithreading_enabled(False)

    from recon_amd import _lib
    from recon_amd.models import SpGAT
    from recon_amd.graph import prepare_graph
    from recon_amd.gat_layers import _fwd_args
    from recon_amd.dist import FlatGradBucket, OverlappedWeightGradSync
    from recon_amd import synth

    workload = args.workload if args.workload != "auto" else ("cfg2" if world == 1 else "cfg4")
    if args.graphs is None:
        args.graphs = 512 if workload == "cfg2" else (1024 if workload == "cfg4" else 64)
    B, n, e, F_, D, H = args.graphs, args.nodes, args.edges, args.feat, args.dim, args.heads
    R = F_
    if workload == "cfg5":
        # BASELINE.json configs[4]: power-law degree graphs of up to 256 nodes / 4 096 edges (SURVEY 8d's generator: n ~ U{16..256},
        # e = min(4096, 16 n), destinations ~ Zipf(1)); `--graphs` graphs per GPU, generated identically on every rank and dealt out by
        # edge count (longest-processing-time packing) so that the ranks' edge loads — what the step's time follows — are level
        import numpy as np
        from recon_amd.dist import shard_by_edges
        rs = np.random.RandomState(0)
        sizes = [int(rs.randint(16, 257)) for _ in range(B * world)]
        ecount = [min(4096, 16 * v) for v in sizes]
        mine = shard_by_edges(ecount, world)[rank]
        dsts, srcs, base = [], [], 0
        for gi in mine:
            rg = np.random.RandomState(1000 + gi)
            p = 1.0 / np.arange(1, sizes[gi] + 1)
            p /= p.sum()
            dsts.append(rg.choice(sizes[gi], size=ecount[gi], p=p) + base)
            srcs.append(rg.randint(0, sizes[gi], size=ecount[gi]) + base)
            base += sizes[gi]
        N = base
        edge = torch.from_numpy(np.stack([np.concatenate(dsts), np.concatenate(srcs)])).long()
        E = edge.shape[1]
        gx = torch.Generator().manual_seed(rank)
        x, ee = torch.randn(N, F_, generator=gx), torch.randn(E, R, generator=gx) * 0.5
        E_global = sum(ecount)
        desc = "cfg5: power-law graphs (<= 256 nodes / 4096 edges each), %d graphs over %d GPU(s) dealt by edge count, H-head KB-GAT attention stage fwd+bwd" % (B * world, world)
    else:
        N, E = B * n, B * e
        x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, R, seed=rank)   # this rank's own graphs
        E_global = world * E
        desc = ("cfg2: H-head KB-GAT attention stage fwd+bwd (heads only, dropout 0)" if workload == "cfg2" else
                "cfg4: BASELINE.json configs[3] — 8-head KB-GAT attention stage fwd+bwd, %d graphs sharded over %d GPU(s) (%d per GPU), "
                "RCCL grad all-reduce; unmeasured on multi-GPU hardware by the builder" % (B * world, world, B))
    torch.manual_seed(0)                                                      # identical parameters on every rank
    model = SpGAT(N, F_, D, R, dropout=0.0, alpha=0.2, nheads=H)
    head_params = [p for att in model.attentions for p in (att.a, att.a_2)]
    cpu_state = {k: v.clone() for k, v in model.state_dict().items()}
    model.to(dev)
    xd = x.to(dev).requires_grad_(True)
    eed = ee.to(dev).requires_grad_(True)
    edged = edge.to(dev)
    nohop = torch.tensor([])
    Gd = torch.randn(N, H * D, generator=torch.Generator().manual_seed(1)).to(dev)
    head_params = model.head_parameters()                                     # every a, then every a_2: the fused gradients [H, D, W] / [H, D] of the
    bucket = FlatGradBucket(head_params)                                      # heads' backward are then two contiguous pieces of the flat buffer,
    model.write_head_gradients_into(bucket)                                   # written in place (no pack copy before the all-reduce)
    graph = prepare_graph(edged, nohop, N)                                    # CSR built once per batch (cached)

    # N > 1: the weight gradient's big term travels (RCCL all-reduce, asynchronous) under the backward's edge chain and comes back from
    # autograd already averaged (recon_amd/dist.py::OverlappedWeightGradSync); RECON_DP_OVERLAP=0 runs the plain schedule — one flat
    # all-reduce after the whole backward — for comparison
    sync = OverlappedWeightGradSync()
    # default: on with RCCL (stream-asynchronous collectives); off with the gloo diagnostic backend, whose device-tensor all-reduce
    # blocks the host at its start (measured with 2 processes on one GPU: 2.42 ms overlapped against 1.99 ms plain)
    want = os.environ.get("RECON_DP_OVERLAP", "1" if (backend == "nccl") else "0")
    overlap = sync.active() and want != "0"

    def step():
        bucket.zero()
        xd.grad = None
        eed.grad = None
        out = model.heads_forward(xd, edged, eed, nohop, nohop)
        if overlap:
            with sync.installed():
                out.backward(Gd)
            bucket.pack()
        else:
            out.backward(Gd)
            bucket.allreduce_mean()

    def step_no_comm():                                                       # the same step, every collective skipped (the plain single-GPU schedule)
        bucket.zero()
        xd.grad = None
        eed.grad = None
        model.heads_forward(xd, edged, eed, nohop, nohop).backward(Gd)
        bucket.pack()

    def barrier_sync():
        if backend is not None:
            dist.barrier()
        torch.cuda.synchronize()

    def reduce_max(v):
        if backend is None:
            return v
        t = torch.tensor([v], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    dt_no_comm = None
    if backend is not None:
        # N > 1: the per-GPU base of the scaling curve, measured HERE — same process, same shard, same clocks — so that value(N) / (N * this)
        # is communication + straggler cost alone, whatever the shard size does to a GPU's own rate (1 024-graph shards run
        # ~8 % more edges/s per GPU than the 512-graph N = 1 workload)
        for _ in range(args.warmup):
            step_no_comm()
        dt_no_comm = timed_region(step_no_comm, args.steps, barrier_sync, reduce_max)
    for _ in range(args.warmup):
        step()
    dt = timed_region(step, args.steps, barrier_sync, reduce_max)
    edges_per_s = E_global * args.steps / dt

    result = {
        "metric": "edges aggregated/sec (GAT fwd+bwd) on synthetic KG-context graphs",
        "value": edges_per_s, "unit": "edges/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": 1e3 * dt / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f32", "data": "synthetic",      # fp32 in, fp32 out, fp32 accumulation; the GEMMs run fp32 operands as 2 f16 (or 3 bf16) terms each
        "config": {"workload": desc,
                   "graphs_per_gpu": B, "nodes_per_graph": n if workload != "cfg5" else "16..256", "edges_per_graph": e if workload != "cfg5" else "min(4096, 16 n)", "F": F_, "R": R,
                   "D_per_head": D, "heads": H, "N_per_gpu": N, "E_per_gpu": E,
                   "parallelism": "dp%d (whole graphs sharded, %s)" % (world, "weight-gradient all-reduce overlapped with the backward's edge chain" if overlap
                                                                         else "flat grad all-reduce after the backward"),
                   "backend": ({"nccl": "nccl (RCCL)"}.get(backend, backend) + (", %d ranks on %d GPU(s): schedule exercised, nothing measured" % (world, ndev) if ndev < world else "")
                               + (", RECON_DIST_FORCE=1: every collective executed at world size 1" if (force and world == 1) else "")) if backend else None},
    }

    if dt_no_comm is not None:
        result.update(scaling_fields(E_global / world, args.steps, dt, dt_no_comm))
        result["scaling_base"] = ("per_gpu_no_comm_*: the same %d steps of the same shard with every collective skipped, timed in this process before "
                                  "the measured loop (max over ranks); efficiency_vs_no_comm = that time / the measured time.  The N = 1 line's "
                                  "secondary.cfg4_shard_1gpu is the same shard size on one GPU in its own process" % args.steps)

    if rank == 0:
        # ---- roofline of the HBM-bound forward edge aggregation (K1', k_gat_atp_fwd: the kernel SURVEY 8d names),
        # timed in situ with HIP events on the launch stream between the score stage and the projection GEMM;
        # the MFMA-bound projection GEMM (the largest single kernel by time) is reported beside it.
        from recon_amd.gat_layers import _atp_args, gat_path_for
        L = _lib.lib()
        f32 = dict(dtype=torch.float32, device=dev)
        with torch.no_grad():
            a = torch.stack([att.a for att in model.attentions]).contiguous()
            a2 = torch.cat([att.a_2 for att in model.attentions], dim=0).contiguous()
        W = 2 * F_ + R
        path = gat_path_for(N, E, F_, R, D, H)
        result["config"]["formulation"] = "aggregate-then-project" if path == "atp" else "project-then-aggregate"
        st = _lib.current_stream()
        ev = [[torch.cuda.Event(enable_timing=True) for _ in range(4)] for _ in range(args.steps)]
        if path == "atp":
            out = torch.empty(N, H * D, **f32)
            u = torch.empty(H, W, **f32)
            c_node = torch.empty(N, 2 * H, **f32)
            c_rel = torch.empty(E, H, **f32)
            V = torch.empty(N, H, W, **f32)
            sigma = torch.empty(E, H, **f32)
            Z = torch.empty(N, H, **f32)
            Zk = torch.empty(N, H, **f32)
            from recon_amd.gat_layers import _atp_split_buffer
            a_split, aux = _atp_split_buffer(F_, R, D, H, dev, N)
            fa = _atp_args(graph, xd.detach(), eed.detach(), a, a2, None, u, c_node, c_rel, V, sigma, Z, Zk, out, 0.2, True, a_split, aux)
            stages = (L.recon_gat_atp_scores, L.recon_gat_atp_aggregate, L.recon_gat_atp_project)
            # compulsory traffic of the aggregation kernel: x and edge_embed rows once, score terms, CSR, V out,
            # saved sigma / Z / Zk  (fp32 values, int32 indices)
            # V: without attention dropout (this workload) and in f16 x 2 mode the destination part x_i Zk/Z is one row for all heads and is
            # written once (for head 0; the GEMMs read that copy): N * (H * (F + R) + F) elements instead of N * H * W
            dst_shared = aux is not None and H > 1 and F_ % 8 == 0
            v_elems = N * (H * (F_ + R) + F_) if dst_shared else N * H * W
            bytes_alg = 4 * (N * F_ + E * R + 2 * N * H + E * H + (N + 1) + 2 * E + v_elems + E * H + 2 * N * H)
            flops_proj = 2.0 * N * W * H * D
            kname, gname = "k_gat_atp_fwd", "k_gemm_f32 (batched projection out = act(V a^T))"
        else:
            P = torch.empty(2, H, N, D, **f32)
            Q = torch.empty(H, E, D, **f32)
            sigma = torch.empty(H, E, **f32)
            Z = torch.empty(H, N, **f32)
            out = torch.empty(N, H * D, **f32)
            fa = _fwd_args(graph, xd.detach(), eed.detach(), a, a2, None, P, Q, sigma, Z, out, 0.2, True)
            stages = (None, L.recon_gat_edge_fwd, L.recon_gat_project)
            bytes_alg = algorithmic_bytes_fwd(N, E, H, D)
            flops_proj = 2.0 * H * D * (2.0 * N * F_ + 1.0 * E * R)
            kname, gname = "k_gat_edge_fwd", "k_gemm_f32 (projections P, Q)"
            a_split = aux = None
        order = (0, 1, 2) if path == "atp" else (2, 1)                 # proj: GEMMs first, then the edge kernel
        for i in range(args.warmup + args.steps):
            k = i - args.warmup
            for pos, si in enumerate(order):
                if stages[si] is None:
                    continue
                if k >= 0:
                    ev[k][pos].record()
                _lib.check(stages[si](C.byref(graph.c), C.byref(fa), st), "stage")
            if k >= 0:
                ev[k][len(order)].record()
        torch.cuda.synchronize()
        ie, ig = order.index(1), order.index(2)
        t_edge = sum(ev[k][ie].elapsed_time(ev[k][ie + 1]) for k in range(args.steps)) / args.steps * 1e-3
        t_proj = sum(ev[k][ig].elapsed_time(ev[k][ig + 1]) for k in range(args.steps)) / args.steps * 1e-3
        result["roofline"] = {"kernel": kname, "bound": "hbm", "achieved": bytes_alg / t_edge / 1e9,
                              "peak": HBM_PEAK / 1e9, "unit": "GB/s", "frac": bytes_alg / t_edge / HBM_PEAK,
                              "traffic": None, "algorithmic_bytes": bytes_alg, "avg_us": t_edge * 1e6,
                              "survey_model_bytes": algorithmic_bytes_fwd(N, E, H, D),
                              "algorithmic_bytes_is": "the builder's byte model of the restructured kernel (inputs, CSR, V as written, saved scores), not SURVEY 8d's B_G",
                              "note": "algorithmic_bytes = compulsory traffic of this kernel as it runs (without attention dropout the "
                                      "destination part of V is written once, not once per head); survey_model_bytes = SURVEY 8d's "
                                      "B_G for the project-then-aggregate layout this kernel no longer needs"}
        if path == "atp":
            # SURVEY 8d's own definition, for reference: B_G over the time of everything in the layer forward except the projection GEMM
            # (here: the score stage and the aggregation kernel), against the same 8 TB/s
            t_scores = sum(ev[k][0].elapsed_time(ev[k][1]) for k in range(args.steps)) / args.steps * 1e-3
            result["roofline"]["survey_definition"] = {"bytes": algorithmic_bytes_fwd(N, E, H, D), "time_us": (t_scores + t_edge) * 1e6,
                                                       "frac": algorithmic_bytes_fwd(N, E, H, D) / (t_scores + t_edge) / HBM_PEAK}
        try:            # HBM traffic of the same kernel from the committed rocprofv3 PMC passes (bench.py cannot collect PMC itself)
            pmc_file = next(f for f in ("round6_pmc_traffic.json", "round5_pmc_traffic.json", "round4_pmc_traffic.json", "round3_pmc_traffic.json", "round2_pmc_traffic.json", "round1_pmc_traffic.json")
                            if os.path.exists(os.path.join(ROOT, "profiles", f)))
            pmc = json.load(open(os.path.join(ROOT, "profiles", pmc_file)))
            if workload == "cfg2" and (B, n, e, F_, D, H) == (512, 16, 64, 200, 200, 8) and kname in pmc:
                result["roofline"]["traffic"] = pmc[kname]["total_bytes"]
                result["roofline"]["traffic_is_from_profiles"] = True
                result["roofline"]["traffic_source"] = ("profiles/%s: rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE per launch of this command, collected "
                                                        "by tools/profile_bench.sh and committed — NOT observed by this run (bench.py cannot collect PMC)") % pmc_file
        except Exception:
            pass
        bx3 = path == "atp" and a_split is not None and aux is None
        if path == "atp" and aux is not None:     # f16 x 2: priced in f16 MFMA flops actually issued (3 term products per fp32 product)
            result["roofline_gemm"] = {"kernel": "k_gemm_hx2 (batched projection out = act(V a^T), 2 x f16 pre-split operands under per-tensor "
                                                 "power-of-two scales, fp32 accumulate)",
                                       "bound": "mfma", "achieved": 3.0 * flops_proj / t_proj / 1e12, "peak": MFMA_BF16_PEAK / 1e12,
                                       "unit": "TFLOP/s", "frac": 3.0 * flops_proj / t_proj / MFMA_BF16_PEAK, "avg_us": t_proj * 1e6,
                                       "fp32_equivalent_tflops": flops_proj / t_proj / 1e12}
        elif bx3:     # priced in bf16 MFMA flops actually issued (6 term products per fp32 product) against the dense bf16 peak
            result["roofline_gemm"] = {"kernel": "k_gemm_bx3 (batched projection out = act(V a^T), 3 x bf16 split operands, fp32 accumulate)",
                                       "bound": "mfma", "achieved": 6.0 * flops_proj / t_proj / 1e12, "peak": MFMA_BF16_PEAK / 1e12,
                                       "unit": "TFLOP/s", "frac": 6.0 * flops_proj / t_proj / MFMA_BF16_PEAK, "avg_us": t_proj * 1e6,
                                       "fp32_equivalent_tflops": flops_proj / t_proj / 1e12}
        else:
            result["roofline_gemm"] = {"kernel": gname, "bound": "mfma", "achieved": flops_proj / t_proj / 1e12,
                                       "peak": MFMA_F32_PEAK / 1e12, "unit": "TFLOP/s", "frac": flops_proj / t_proj / MFMA_F32_PEAK,
                                       "avg_us": t_proj * 1e6}

        # ---- the same step on the exact-fp32 MFMA GEMMs (RECON_GEMM_BX3=0: v_mfma_f32_32x32x2_f32, no split operands): brackets the
        # "dtype f32" claim of the main line, whose three large products run fp32 operands as two f16 terms each
        if world == 1 and not args.no_extras:
            from recon_amd import gat_layers
            from recon_amd.graph import clear_graph_cache
            n_x = max(10, args.steps // 4)

            def timed(fn, k):
                for _ in range(3):
                    fn()
                torch.cuda.synchronize()
                tq = time.perf_counter()
                for _ in range(k):
                    fn()
                torch.cuda.synchronize()
                return (time.perf_counter() - tq) / k * 1e3
            fam = gat_layers._GEMM_BX3
            gat_layers._GEMM_BX3 = "0"
            try:
                ms_x = timed(step, n_x)
            finally:
                gat_layers._GEMM_BX3 = fam
            result["exact_fp32"] = {"ms_per_step": ms_x, "edges_per_s": E / ms_x * 1e3, "steps": n_x,
                                    "what": "same step, every GEMM on the exact fp32 matrix-core path (gat_layers._GEMM_BX3 = '0')"}
            # ---- SURVEY 8d's "uncached" figure: a NEW edge tensor every step (as every iteration of the reference's loops brings one), so
            # the CSR / CSC build (2 radix sorts, row pointers, hub scan, one host sync) is inside the step
            fresh = [edged.clone() for _ in range(n_x + 3)]
            it = iter(fresh)

            def step_uncached():
                ed = next(it)
                bucket.zero()
                xd.grad = None
                eed.grad = None
                model.heads_forward(xd, ed, eed, nohop, nohop).backward(Gd)
            for _ in range(3):
                step_uncached()
            torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(n_x):
                step_uncached()
            torch.cuda.synchronize()
            ms_u = (time.perf_counter() - tq) / n_x * 1e3
            clear_graph_cache()
            result["uncached"] = {"ms_per_step": ms_u, "edges_per_s": E / ms_u * 1e3, "steps": n_x,
                                  "what": "same step with the graph preparation (COO -> CSR + CSC, hub tables) rebuilt from a fresh edge tensor every step"}
            # the same with edge tensors the producer vouches for (graph.trust: what recon_amd.sampler hands out) — no range check, one host read less per build
            from recon_amd.graph import trust
            fresh = [trust(edged.clone(), bound=N) for _ in range(n_x + 3)]
            it = iter(fresh)
            for _ in range(3):
                step_uncached()
            torch.cuda.synchronize()
            tq = time.perf_counter()
            for _ in range(n_x):
                step_uncached()
            torch.cuda.synchronize()
            ms_t = (time.perf_counter() - tq) / n_x * 1e3
            clear_graph_cache()
            result["uncached"]["trusted_edges_ms_per_step"] = ms_t
            # ---- the other BASELINE.json configurations that fit one GPU (SURVEY 8d's cfg 3a / 3b / 5), priced with 8d's own formulas
            try:
                sys.path.insert(0, os.path.join(ROOT, "tools"))
                import secondary
                result["secondary"] = secondary.all_secondary(fast=True)
            except Exception as exc:                                    # never lose the main line to a secondary workload
                result["secondary"] = {"error": repr(exc)[:300]}

        # ---- CPU baseline: the oracle issuing the reference's own ATen op sequence (sparse_coo_tensor ->
        # sparse.sum -> to_dense), all host cores, same workload, bounded to ~args.cpu_seconds.
        if world == 1 and not args.no_cpu_baseline:
            from oracle import recon_oracle as O          # the checker, timed here as the baseline only
            heads_a = [cpu_state["attention_%d.a" % h].clone().requires_grad_(True) for h in range(H)]
            heads_a2 = [cpu_state["attention_%d.a_2" % h].clone().requires_grad_(True) for h in range(H)]
            xc, eec, Gc = x.clone().requires_grad_(True), ee.clone().requires_grad_(True), Gd.cpu()

            def cpu_step(nh=H):
                outs = [O.gat_layer_forward(xc, edge, eec, None, None, heads_a[h], heads_a2[h], 0.2, True,
                                            aten_sequence=True) for h in range(nh)]
                torch.cat(outs, dim=1).backward(Gc[:, :nh * D])
            # pick the thread count that runs this op sequence fastest on this host (one-head probe);
            # all-core runs of these small sparse ops are far slower than 8-32 threads
            ncpu = os.cpu_count() or 1
            best = None
            for th in sorted({t for t in (4, 8, 16, 32, 64) if t <= ncpu} | {min(ncpu, 8)}):
                torch.set_num_threads(th)
                cpu_step(1)
                tq = time.perf_counter()
                cpu_step(1)
                tq = time.perf_counter() - tq
                if best is None or tq < best[0]:
                    best = (tq, th)
            cores = best[1]
            torch.set_num_threads(cores)
            cpu_step()
            n_cpu, t_cpu0 = 0, time.perf_counter()
            while n_cpu < 10 and (time.perf_counter() - t_cpu0) < args.cpu_seconds:
                cpu_step()
                n_cpu += 1
            t_cpu = (time.perf_counter() - t_cpu0) / n_cpu
            result["cpu_baseline"] = {"value": E / t_cpu, "unit": "edges/s", "cores": cores, "kind": "port",
                                      "sample": "%d full cfg2 steps (same batch, %d heads, fwd+bwd) after 1 warm-up, "
                                                "oracle with the reference's ATen op sequence, torch %d threads "
                                                "(fastest of 4..64 on a %d-cpu host)" % (n_cpu, H, cores, ncpu),
                                      "ms_per_step": 1e3 * t_cpu}
            # ---- stock PyTorch-ROCm eager on the SAME GPU: the same ATen op sequence with device tensors,
            # i.e. what the unmodified reference would run as on this machine.  Reported, never the target.
            try:
                ga = [t.detach().to(dev).requires_grad_(True) for t in heads_a]
                ga2 = [t.detach().to(dev).requires_grad_(True) for t in heads_a2]
                xg, eeg = xd.detach().clone().requires_grad_(True), eed.detach().clone().requires_grad_(True)

                def eager_step():
                    outs = [O.gat_layer_forward(xg, edged, eeg, None, None, ga[h], ga2[h], 0.2, True, aten_sequence=True)
                            for h in range(H)]
                    torch.cat(outs, dim=1).backward(Gd)
                eager_step()
                torch.cuda.synchronize(dev)
                t_e0 = time.perf_counter()
                for _ in range(3):
                    eager_step()
                torch.cuda.synchronize(dev)
                t_e = (time.perf_counter() - t_e0) / 3
                result["eager_rocm_baseline"] = {"value": E / t_e, "unit": "edges/s", "ms_per_step": 1e3 * t_e, "kind": "port",
                                                 "sample": "3 full cfg2 steps after 1 warm-up, the oracle's ATen op sequence "
                                                           "(sparse_coo_tensor -> sparse.sum -> to_dense, %d heads in a Python "
                                                           "loop) executed by PyTorch-ROCm eager on cuda:0" % H}
            except Exception as exc:                                   # an op of the sequence unsupported by this build
                result["eager_rocm_baseline"] = {"value": None, "error": repr(exc)[:200]}
        print(json.dumps(result))
    if backend is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
