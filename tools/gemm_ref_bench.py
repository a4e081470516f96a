#!/usr/bin/env python3
"""Calibration for the f16 x 2 GEMMs of the layer (cfg 2): what do the vendor's tuned f16 kernels (hipBLASLt through torch) need
for the SAME matrix-core work — the three term products of one split GEMM are a plain f16 GEMM with K tripled — and what does
a plain streaming pass over the same bytes cost.  Not a product path; numbers go to DESIGN.md."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def time_once(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) * 1e3 / iters


def med(fn, rounds=5, iters=10):
    fn(); torch.cuda.synchronize()
    ts = sorted(time_once(fn, iters) for _ in range(rounds))
    return ts[len(ts) // 2]


def main():
    d = torch.device("cuda:0")
    H, N, W, D = 8, 8192, 600, 200
    flush = torch.empty(512 << 20, dtype=torch.uint8, device=d)
    for name, (b, m, n, k) in {"projection (b=8, 8192 x 200, K=3*600)": (H, N, D, 3 * W),
                               "projection flat (65536 x 200, K=3*600)": (1, H * N, D, 3 * W),
                               "g_V (b=8, 8192 x 600, K=3*200)": (H, N, W, 3 * D),
                               "weights (b=8, 600 x 200, K=3*8192)": (H, W, D, 3 * N)}.items():
        A = torch.randn(b, m, k, device=d, dtype=torch.float16)
        B = torch.randn(b, k, n, device=d, dtype=torch.float16)
        out = torch.empty(b, m, n, device=d, dtype=torch.float16)
        t = med(lambda: torch.bmm(A, B, out=out))
        Bt = torch.randn(b, n, k, device=d, dtype=torch.float16)
        t2 = med(lambda: torch.bmm(A, Bt.transpose(1, 2), out=out))
        fl = 2.0 * b * m * n * k
        print("%-46s  NN %7.1f us %6.0f TF   NT %7.1f us %6.0f TF   (f16 out)" % (name, t, fl / t / 1e6, t2, fl / t2 / 1e6), flush=True)
        if name.startswith("weights"):
            At = torch.randn(b, k, m, device=d, dtype=torch.float16)
            t3 = med(lambda: torch.bmm(At.transpose(1, 2), B, out=out))
            print("%-46s  TN %7.1f us %6.0f TF" % ("", t3, fl / t3 / 1e6), flush=True)
    # streaming passes of the GEMMs' compulsory bytes
    for name, (rd, wr) in {"projection bytes (157 MB in, 52 MB out)": (157, 52), "g_V bytes (52 MB in, 157 MB out)": (52, 157),
                           "K2' bytes (199 in, 61 out)": (199, 61)}.items():
        src = torch.empty(rd << 20, dtype=torch.uint8, device=d).view(torch.float32)
        dst = torch.empty(wr << 20, dtype=torch.uint8, device=d).view(torch.float32)
        n = min(src.numel(), dst.numel())
        # read rd, write wr: copy the common part, then reduce / fill the remainder
        def f():
            dst[:n].copy_(src[:n])
            if src.numel() > n:
                src[n:].sum()
            else:
                dst[n:].fill_(1.0)
        t = med(f)
        print("%-46s  %7.1f us  %5.2f TB/s (copy + sum/fill, three launches)" % (name, t, (rd + wr) * 1.048576 / t), flush=True)
    del flush


if __name__ == "__main__":
    main()
