#!/usr/bin/env python3
"""The driver's bench protocol, repeated: `python bench.py --steps 20 --warmup 5` as REPS fresh processes (ms_per_step of each), then one
process that times every one of its first steps with HIP events (no host sync inside) — how long a fresh process takes to reach the
200-step figure.  Runs on the GPU box:  python3 tools/driver_protocol.py [REPS]"""
import json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def per_step_trace(nsteps=120):
    import torch
    torch.autograd.set_multithreading_enabled(False)
    from recon_amd.models import SpGAT
    from recon_amd.graph import prepare_graph
    from recon_amd.dist import FlatGradBucket
    from recon_amd import synth
    dev = torch.device("cuda:0")
    B, n, e, F_, D, H = 512, 16, 64, 200, 200, 8
    N = B * n
    x, edge, ee = synth.synthetic_batched_graph(B, n, e, F_, F_, seed=0)
    torch.manual_seed(0)
    model = SpGAT(N, F_, D, F_, dropout=0.0, alpha=0.2, nheads=H).to(dev)
    xd, eed, edged = x.to(dev).requires_grad_(True), ee.to(dev).requires_grad_(True), edge.to(dev)
    nohop = torch.tensor([])
    Gd = torch.randn(N, H * D, generator=torch.Generator().manual_seed(1)).to(dev)
    bucket = FlatGradBucket(model.head_parameters())
    model.write_head_gradients_into(bucket)
    prepare_graph(edged, nohop, N)

    def step():
        bucket.zero(); xd.grad = None; eed.grad = None
        model.heads_forward(xd, edged, eed, nohop, nohop).backward(Gd)
        bucket.allreduce_mean()
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)]
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    ev[0].record()
    host = []
    for i in range(nsteps):
        h0 = time.perf_counter()
        step()
        host.append((time.perf_counter() - h0) * 1e3)
        ev[i + 1].record()
    torch.cuda.synchronize()
    wall = (time.perf_counter() - t0) * 1e3
    dts = [ev[i].elapsed_time(ev[i + 1]) for i in range(nsteps)]
    return {"device_ms_per_step": [round(v, 3) for v in dts], "host_enqueue_ms_per_step": [round(v, 3) for v in host], "wall_ms": wall}


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--trace":
        print(json.dumps(per_step_trace()))
        sys.exit(0)
    reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    out = {"driver_protocol_ms_per_step": [], "steps200_ms_per_step": []}
    for args, key, k in ((["--steps", "20", "--warmup", "5"], "driver_protocol_ms_per_step", reps), (["--steps", "200", "--warmup", "20"], "steps200_ms_per_step", 2)):
        for _ in range(k):
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + args + ["--no-extras", "--no-cpu-baseline"], capture_output=True, text=True)
            line = [l for l in r.stdout.splitlines() if l.startswith("{")]
            out[key].append(json.loads(line[-1])["ms_per_step"] if line else None)
    r = subprocess.run([sys.executable, os.path.abspath(__file__), "--trace"], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("{")]
    out["fresh_process_trace"] = json.loads(line[-1]) if line else r.stderr[-500:]
    print(json.dumps(out))
