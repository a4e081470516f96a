"""Practical HBM rates on this box through torch's own streaming kernels: pure write (fill), pure read (sum), copy — context for the roofline
fractions of write-heavy kernels (k_gat_atp_fwd writes 115 of its 151 MB)."""
import time, torch
d = torch.device("cuda:0")
for mb in (128, 512, 2048):
    n = mb * 1024 * 1024 // 4
    x = torch.empty(n, dtype=torch.float32, device=d); y = torch.empty_like(x)
    def t(f, reps=50):
        for _ in range(5): f()
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps): f()
        torch.cuda.synchronize(); return (time.perf_counter() - t0) / reps
    tw = t(lambda: x.fill_(1.0)); tr = t(lambda: x.sum()); tc = t(lambda: y.copy_(x))
    print("%5d MB: write %.2f TB/s  read %.2f TB/s  copy %.2f TB/s (read + write bytes)" % (mb, mb * 1.048576e6 / tw / 1e12, mb * 1.048576e6 / tr / 1e12, 2 * mb * 1.048576e6 / tc / 1e12))
