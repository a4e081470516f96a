// Stand-alone driver of k_propagate_fwd_h at cfg 3b with s_memtime stamps (-DRECON_PROP_STAMPS): where a hop's cycles go.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -DRECON_PROP_STAMPS tools/probe/prop_h_stamps.hip -o tools/probe/prop_h_stamps
#include "../../recon_amd/csrc/prop_h.hip"
#include <stdio.h>
#include <vector>
using namespace recon;
int main(int argc, char** argv) {
    const int B = 1024, n = 9, d = 8, L = 3, C = n * (n - 1), S = 2 * d * n, dd = 2 * d;
    const bool save = argc > 1;
    std::vector<float> hA(static_cast<size_t>(B) * S * S), hh(static_cast<size_t>(B) * C * S);
    unsigned x = 12345;
    auto rnd = [&]() { x = x * 1664525u + 1013904223u; return (x >> 8) * (1.0f / 16777216.f); };
    for (auto& v : hA) { float r = rnd(); v = r > 0.5f ? 0.1f * (r - 0.5f) : 0.f; }
    for (auto& v : hh) v = rnd() - 0.5f;
    std::vector<int64_t> hd(C * dd), tl(C * dd);
    { int c = 0; for (int i = 0; i < n; ++i) for (int j = 0; j < n; ++j) if (i != j) { for (int k = 0; k < dd; ++k) { hd[c * dd + k] = i * dd + k; tl[c * dd + k] = j * dd + k; } ++c; } }
    PropK p{};
    float* dA[3];
    for (int l = 0; l < L; ++l) { hipMalloc(&dA[l], hA.size() * 4); hipMemcpy(dA[l], hA.data(), hA.size() * 4, hipMemcpyHostToDevice); p.adj[l] = dA[l]; }
    float *dh, *dout, *dsave = nullptr; int64_t *dhd, *dtl;
    hipMalloc(&dh, hh.size() * 4); hipMemcpy(dh, hh.data(), hh.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dout, static_cast<size_t>(B) * C * L * dd * 4);
    if (save) hipMalloc(&dsave, static_cast<size_t>(L) * B * C * S * 4);
    hipMalloc(&dhd, hd.size() * 8); hipMalloc(&dtl, tl.size() * 8);
    hipMemcpy(dhd, hd.data(), hd.size() * 8, hipMemcpyHostToDevice); hipMemcpy(dtl, tl.data(), tl.size() * 8, hipMemcpyHostToDevice);
    p.h0 = dh; p.h0_bs = static_cast<int64_t>(C) * S; p.head = dhd; p.tail = dtl; p.idx_bs = 0; p.out = dout; p.hsave = dsave;
    p.B = B; p.C = C; p.S = S; p.L = L; p.dd = dd; p.act = RECON_ACT_RELU; p.Sp = S; p.pitch = S + 4; p.CC = 0;
    if (save) { float* dstat; hipMalloc(&dstat, static_cast<size_t>(B) * (2 * L + 1) * 4); p.stats = dstat; }
    unsigned long long* dst; const size_t ns = 2 * 2 * 4 * 8 * 16;
    hipMalloc(&dst, ns * 8); hipMemset(dst, 0, ns * 8);
#ifdef RECON_PROP_STAMPS
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &dst, sizeof(dst));
#endif
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 3; ++it) prop_fwd_h(p, 0);
    hipEventRecord(e0); for (int it = 0; it < 20; ++it) prop_fwd_h(p, 0); hipEventRecord(e1); hipDeviceSynchronize();
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("forward%s: %.1f us per launch\n", save ? " (states saved)" : "", ms / 20 * 1e3);
#ifdef RECON_PROP_STAMPS
    const bool show = true;
#else
    const bool show = false;
#endif
    std::vector<unsigned long long> st(ns);
    hipMemcpy(st.data(), dst, ns * 8, hipMemcpyDeviceToHost);
    if (show) {
    const char* names[9] = {"hop start", "rows split", "prefetch out", "products", "epilogue 1", "barrier 1", "epilogue 2", "barrier 2", "gather/save"};
    for (int blk = 0; blk < 2; ++blk) for (int w = 0; w < 2; ++w) {
        printf("block %d wave %s (cycles since the graph's start; delta)\n", blk ? 77 : 0, w ? "last" : "0");
        for (int g = 0; g < 4; ++g) {
            auto at = [&](int hop, int slot) { return st[(((blk * 2 + w) * 4 + g) * 8 + hop) * 16 + slot]; };
            const unsigned long long t0 = at(7, 0);
            printf("  graph %d: staged +%llu, barrier +%llu |", g, at(7, 1) - t0, at(7, 2) - t0);
            for (int l = 0; l < L; ++l) {
                printf(" hop %d:", l);
                for (int s = 1; s < 9; ++s) printf(" %llu", at(l, s) - at(l, s - 1));
                printf(" |");
            }
            printf(" end +%llu\n", at(6, 0) - t0);
        }
    }
    printf("columns per hop: "); for (int s = 1; s < 9; ++s) printf("%s, ", names[s]); printf("\n");
    }
    if (!save) return 0;
    // ---- backward on the saved states
    PropBwdH q{};
    float* dgA[3]; float *dgout, *dgH;
    for (int l = 0; l < L; ++l) { hipMalloc(&dgA[l], hA.size() * 4); q.adj[l] = dA[l]; q.gadj[l] = dgA[l]; }
    hipMalloc(&dgout, static_cast<size_t>(B) * C * L * dd * 4); hipMemcpy(dgout, dout, static_cast<size_t>(B) * C * L * dd * 4, hipMemcpyDeviceToDevice);
    hipMalloc(&dgH, hh.size() * 4);
    q.h0 = dh; q.h0_bs = p.h0_bs; q.hsave = dsave; q.head = dhd; q.tail = dtl; q.idx_bs = 0; q.gout = dgout; q.gH = dgH; q.stats = p.stats;
    q.B = B; q.C = C; q.S = S; q.L = L; q.dd = dd; q.act = RECON_ACT_RELU;
    hipMemset(dst, 0, ns * 8);
    for (int it = 0; it < 3; ++it) prop_bwd_h(q, 0);
    hipEventRecord(e0); for (int it = 0; it < 20; ++it) prop_bwd_h(q, 0); hipEventRecord(e1); hipDeviceSynchronize();
    hipEventElapsedTime(&ms, e0, e1);
    printf("backward (3 hops): %.1f us per launch\n", ms / 20 * 1e3);
    hipMemcpy(st.data(), dst, ns * 8, hipMemcpyDeviceToHost);
    if (show) for (int blk = 0; blk < 2; ++blk) for (int w = 0; w < 2; ++w) {
        printf("bwd block %d wave %s\n", blk ? 77 : 0, w ? "last" : "0");
        for (int g = 0; g < 4; ++g) {
            auto at = [&](int hop, int slot) { return st[(((blk * 2 + w) * 4 + g) * 8 + hop) * 16 + slot]; };
            const unsigned long long t0 = at(7, 0);
            printf("  graph %d: first Y +%llu |", g, at(7, 1) - t0);
            for (int h = 0; h < L; ++h) {
                printf(" hop %d: (c) %llu [steps", h, at(h, 1) - at(h, 0));
                for (int k = 0; k < 3; ++k) printf(" %llu", at(h, 5 + k) - at(h, 0));
                printf("] store %llu (d) %llu [steps", at(h, 2) - at(h, 1), at(h, 3) - at(h, 2));
                for (int k = 0; k < 5; ++k) printf(" %llu", at(h, 8 + k) - at(h, 2));
                printf(" | step 2: pre-barrier +%llu barrier %llu store %llu rest-to-stamp %llu", at(h, 13) - at(h, 9), at(h, 14) - at(h, 13), at(h, 15) - at(h, 14), at(h, 10) - at(h, 15));
                if (h + 1 < L) printf("] make_y %llu |", at(h, 4) - at(h, 3)); else printf("] |");
            }
            printf("\n");
        }
    }
    return 0;
}
