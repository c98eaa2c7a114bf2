// kbench -- micro-benchmarks of single launches through the C ABI, without Python / torch (a fresh GPU box spends
// 1-2 minutes on its first `import torch`; this starts in a second).  Tuning aid, not part of the product.
//
//   kbench conv <shape> <impl> [B=256] [iters=200] [ref_impl=-1]   time one fused conv; ref_impl >= 0: max|diff| against that kernel
//   kbench rvq [rows=256] [iters=200]                               time adk_rvq_encode; prints an index checksum (compare ADK_RVQ_V1=1 / ADK_RVQ_V=2)
//
// shapes: see kShapes below (the layers of the vctk_v1 pipeline at one frame per stream).
// Build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -I include tools/kbench.cpp -L audiodec_amd -laudiodec_hip -Wl,-rpath,'$ORIGIN/../../audiodec_amd' -o tools/bin/kbench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <dlfcn.h>
#include <cstdlib>
#include <cstring>
#include <cmath>
#include <algorithm>
#include <string>
#include <vector>
#include <random>
#include "audiodec_hip.h"

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)
#define AK(x) do { int r_ = (x); if (r_ != 0) { fprintf(stderr, "%s: %s\n", #x, adk_last_error()); exit(3); } } while (0)

struct Shape { const char* name; int cin_g, cout_g, groups, taps, stride, dil, t_out, up, act, res; };
static const Shape kShapes[] = {
    // vocoder grouped K11 convs (stage 0..3), dilation 5 and 1
    {"s0", 256, 256, 3, 11, 1, 5, 5, 1, 2, 0},   {"s0d1", 256, 256, 3, 11, 1, 1, 5, 1, 2, 1},
    {"s1", 128, 128, 3, 11, 1, 5, 25, 1, 2, 0},  {"s1d1", 128, 128, 3, 11, 1, 1, 25, 1, 2, 1},
    {"s2", 64, 64, 3, 11, 1, 5, 100, 1, 2, 0},   {"s2d1", 64, 64, 3, 11, 1, 1, 100, 1, 2, 1},
    {"s3", 32, 32, 3, 11, 1, 5, 300, 1, 2, 0},   {"s3d1", 32, 32, 3, 11, 1, 1, 300, 1, 2, 1},
    // encoder residual-unit K7 convs (block 3..0) and their 1x1 + residual
    {"e3", 256, 256, 1, 7, 1, 9, 5, 1, 1, 0},    {"e2", 128, 128, 1, 7, 1, 9, 25, 1, 1, 0},
    {"e1", 64, 64, 1, 7, 1, 9, 100, 1, 1, 0},    {"e0", 32, 32, 1, 7, 1, 9, 300, 1, 1, 0},
    {"r3", 256, 256, 1, 1, 1, 1, 5, 1, 1, 1},    {"r2", 128, 128, 1, 1, 1, 1, 25, 1, 1, 1},
    {"r1", 64, 64, 1, 1, 1, 1, 100, 1, 1, 1},    {"r0", 32, 32, 1, 1, 1, 1, 300, 1, 1, 1},
    // strided encoder convs, projector, vocoder input conv
    {"d0", 32, 64, 1, 6, 3, 1, 100, 1, 0, 0},    {"d1", 64, 128, 1, 8, 4, 1, 25, 1, 0, 0},
    {"d2", 128, 256, 1, 10, 5, 1, 5, 1, 0, 0},   {"d3", 256, 512, 1, 10, 5, 1, 1, 1, 0, 0},
    {"p", 512, 64, 1, 3, 1, 1, 1, 1, 0, 0},      {"in", 64, 512, 1, 7, 1, 1, 1, 1, 0, 0},
    // transposed convs in polyphase form (up = stride), 1x1 fusion convs
    {"up0", 512, 1280, 1, 2, 1, 1, 1, 5, 2, 0},  {"up1", 256, 640, 1, 2, 1, 1, 5, 5, 2, 0},
    {"up2", 128, 256, 1, 2, 1, 1, 25, 4, 2, 0},  {"up3", 64, 96, 1, 2, 1, 1, 100, 3, 2, 0},
    {"o0", 768, 256, 1, 1, 1, 1, 5, 1, 0, 0},    {"o1", 384, 128, 1, 1, 1, 1, 25, 1, 0, 0},
    {"o2", 192, 64, 1, 1, 1, 1, 100, 1, 0, 0},   {"o3", 96, 32, 1, 1, 1, 1, 300, 1, 0, 0},
};

static float* dev_random(size_t n, float scale, unsigned seed) {
    std::vector<float> h(n);
    std::mt19937 g(seed);
    std::normal_distribution<float> d(0.f, 1.f);
    for (auto& v : h) v = d(g) * scale;
    float* p; CK(hipMalloc(&p, n * 4)); CK(hipMemcpy(p, h.data(), n * 4, hipMemcpyHostToDevice));
    return p;
}

static int cmd_conv(int argc, char** argv) {
    if (argc < 4) { fprintf(stderr, "kbench conv <shape> <impl> [B] [iters] [ref_impl]\n"); return 1; }
    const Shape* sh = nullptr;
    for (const Shape& s : kShapes) if (!strcmp(s.name, argv[2])) sh = &s;
    if (!sh) { fprintf(stderr, "unknown shape %s\n", argv[2]); return 1; }
    const int impl = atoi(argv[3]);
    const int B = argc > 4 ? atoi(argv[4]) : 256, iters = argc > 5 ? atoi(argv[5]) : 200, ref_impl = argc > 6 ? atoi(argv[6]) : -1;
    const int hist = (sh->taps - 1) * sh->dil, rows = hist + sh->t_out * sh->stride;
    const int cin_t = sh->cin_g * sh->groups, M = sh->cout_g * sh->groups, cout_real = M / sh->up, ktot = sh->taps * sh->cin_g;
    float* ring = dev_random((size_t)B * rows * cin_t, 1.f, 1);
    float* w = dev_random((size_t)M * ktot, 1.f / std::sqrt((float)ktot), 2);
    float* bias = dev_random(M, 0.1f, 3);
    float* res = sh->res ? dev_random((size_t)B * sh->t_out * M, 1.f, 4) : nullptr;
    const size_t out_n = (size_t)B * sh->t_out * sh->up * cout_real;
    float *out, *out_ref; CK(hipMalloc(&out, out_n * 4)); CK(hipMalloc(&out_ref, out_n * 4));
    auto packed = [&](int im) -> float* {
        const bool s16 = im >= 4;
        const int64_t n = s16 ? adk_packed_weight_floats_split16(sh->groups, sh->cout_g, ktot) : adk_packed_weight_floats(sh->groups, sh->cout_g, ktot);
        if (n < 0) return nullptr;
        float* p; CK(hipMalloc(&p, n * 4));
        if (s16) AK(adk_pack_weights_split16(w, p, sh->groups, sh->cout_g, ktot, nullptr)); else AK(adk_pack_weights_mfma(w, p, sh->groups, sh->cout_g, ktot, nullptr));
        return p;
    };
    adk_conv_desc d; memset(&d, 0, sizeof(d));
    d.cin_g = sh->cin_g; d.cout_g = sh->cout_g; d.groups = sh->groups; d.taps = sh->taps; d.stride = sh->stride; d.dilation = sh->dil;
    d.hist = hist; d.up = sh->up; d.cout_real = cout_real; d.in_group_stride = sh->cin_g; d.res_group_stride = sh->cout_g;
    d.act_in = sh->act; d.act_in_slope = 0.1f; d.act_out = 0; d.w = w; d.bias = bias;
    adk_ring_view vin{ring, rows, cin_t, hist, 0}, vout{out, sh->t_out * sh->up, cout_real, 0, 0}, vres{res, sh->t_out, M, 0, 0};
    if (!res) memset(&vres, 0, sizeof(vres));
    d.w_frag = impl == 1 ? nullptr : packed(impl);
    char name[64] = "";
    AK(adk_causal_conv_describe(&d, vin, vout, vres, B, sh->t_out, impl, name, 64));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int i = 0; i < 5; ++i) AK(adk_causal_conv(&d, vin, vout, vres, B, sh->t_out, impl, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) AK(adk_causal_conv(&d, vin, vout, vres, B, sh->t_out, impl, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = 1e3 * ms / iters;
    const double flops = 2.0 * M * ktot * sh->t_out * B;
    const double bytes = 4.0 * ((double)B * rows * cin_t + (double)M * ktot + (double)out_n * (res ? 2 : 1));
    printf("conv %-5s impl %d %-18s B=%d  %8.2f us  %7.1f TF  %6.2f TB/s (algorithmic %.1f MB)", sh->name, impl, name, B, us, flops / us / 1e6, bytes / us / 1e6, bytes / 1e6);
    if (ref_impl >= 0) {
        adk_ring_view vref = vout; vref.base = out_ref;
        adk_conv_desc d2 = d; d2.w_frag = ref_impl == 1 ? nullptr : packed(ref_impl);
        AK(adk_causal_conv(&d2, vin, vref, vres, B, sh->t_out, ref_impl, st));
        CK(hipStreamSynchronize(st));
        std::vector<float> a(out_n), b(out_n);
        CK(hipMemcpy(a.data(), out, out_n * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), out_ref, out_n * 4, hipMemcpyDeviceToHost));
        double md = 0, mx = 0; size_t nbad = 0;
        for (size_t i = 0; i < out_n; ++i) { double df = std::fabs((double)a[i] - b[i]); if (!(df <= 1e30)) ++nbad; md = std::max(md, df); mx = std::max(mx, (double)std::fabs(b[i])); }
        printf("  max|d| vs impl %d = %.3e (|ref|max %.2f, nonfinite %zu)", ref_impl, md, mx, nbad);
    }
    int32_t flags = 0; AK(adk_debug_flags(&flags));
    {   // FNV-1a over the output bits: two builds / kernels that claim bit-identical results print the same value
        std::vector<unsigned> ob(out_n);
        CK(hipMemcpy(ob.data(), out, out_n * 4, hipMemcpyDeviceToHost));
        unsigned long long h = 1469598103934665603ull;
        for (unsigned v : ob) { h ^= v; h *= 1099511628211ull; }
        printf("  out#%016llx", h);
    }
    printf("  flags %d\n", flags);
    // debug builds of the library (-DADK_SK16_DBG=16) record an s_memtime timeline of one workgroup of the last launch
    typedef int (*trace_fn)(unsigned long long*, int);
    if (trace_fn tf = (trace_fn)dlsym(RTLD_DEFAULT, "adk_debug_sk_trace")) {
        std::vector<unsigned long long> tr(64 * 8);
        if (tf(tr.data(), 64 * 8) == 0) {
            printf("  iteration timeline of one workgroup, wave 0 (s_memtime ticks between stamps: loads issued | first MFMA step | all MFMAs issued | converted + staged | segment end | barrier)\n");
            double sum[7] = {0}; int cnt = 0;
            for (int it = 0; it < 64 && tr[it * 8] != 0; ++it) {
                const unsigned long long* t = &tr[it * 8];
                if (!t[6]) break;
                const double d[6] = {double(t[1] - t[0]), double(t[2] - t[1]), double(t[3] - t[2]), double(t[4] - t[3]), double(t[5] - t[4]), double(t[6] - t[5])};
                const double gap = it > 0 && tr[(it - 1) * 8 + 6] ? double(t[0] - tr[(it - 1) * 8 + 6]) : 0.0;
                if (it < 24) printf("   it %2d: %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f   (+%.0f to the next top)  total %6.0f\n", it, d[0], d[1], d[2], d[3], d[4], d[5], gap, double(t[6] - t[0]));
                if (it >= 1) { for (int k = 0; k < 6; ++k) sum[k] += d[k]; sum[6] += gap; ++cnt; }
            }
            if (cnt) printf("   mean over %d iterations: %6.0f %6.0f %6.0f %6.0f %6.0f %6.0f  gap %.0f\n", cnt, sum[0] / cnt, sum[1] / cnt, sum[2] / cnt, sum[3] / cnt, sum[4] / cnt, sum[5] / cnt, sum[6] / cnt);
        }
    }
    if (trace_fn wf = (trace_fn)dlsym(RTLD_DEFAULT, "adk_debug_sk_wg_trace")) {
        std::vector<unsigned long long> tr(512 * 4);
        if (wf(tr.data(), 512 * 4) == 0) {
            unsigned long long t0 = ~0ull, t3 = 0; int n = 0;
            for (int r = 0; r < 512; ++r) if (tr[r * 4]) { t0 = std::min(t0, tr[r * 4]); t3 = std::max(t3, tr[r * 4 + 3]); ++n; }
            printf("  per-workgroup wall clock of the last launch (10 ns ticks since the first workgroup started; %d workgroups, span %llu):\n", n, t3 - t0);
            printf("   range: start  loop-entry  before-last-iterations  end\n");
            double s0 = 0, s1 = 0, s2 = 0, s3 = 0;
            for (int r = 0; r < 512; ++r) if (tr[r * 4]) {
                if (r % 16 == 0 || tr[r * 4 + 3] == t3) printf("   %3d: %5llu %5llu %5llu %5llu%s\n", r, tr[r * 4] - t0, tr[r * 4 + 1] - t0, tr[r * 4 + 2] - t0, tr[r * 4 + 3] - t0, tr[r * 4 + 3] == t3 ? "  <- last" : "");
                s0 += tr[r * 4] - t0; s1 += tr[r * 4 + 1] - t0; s2 += tr[r * 4 + 2] - t0; s3 += tr[r * 4 + 3] - t0;
            }
            printf("   mean: %5.0f %5.0f %5.0f %5.0f\n", s0 / n, s1 / n, s2 / n, s3 / n);
        }
    }
    if (trace_fn pf = (trace_fn)dlsym(RTLD_DEFAULT, "adk_debug_rp_trace")) {
        std::vector<unsigned long long> tr(2 * 16 * 8);
        if (pf(tr.data(), 2 * 16 * 8) == 0)
            for (int wg = 0; wg < 2; ++wg) {
                printf("  pipelined rows kernel, workgroup %d wave 0, 10 ns ticks per item: k loop | epilogue | left-over staging | barrier\n", wg ? 100 : 0);
                for (int it = 0; it < 16; ++it) {
                    const unsigned long long* t = &tr[(wg * 16 + it) * 8];
                    if (!t[0] || !t[3]) break;
                    printf("   item %2d (t = %5llu): %5llu %5llu %5llu %5lld\n", it, t[0] - tr[wg * 16 * 8], t[1] - t[0], t[2] - t[1], t[3] - t[2], t[4] ? (long long)(t[4] - t[3]) : -1ll);
                }
            }
    }
    if (trace_fn wf = (trace_fn)dlsym(RTLD_DEFAULT, "adk_debug_rl_wg_trace")) {
        std::vector<unsigned long long> tr(2048 * 4);
        if (wf(tr.data(), 2048 * 4) == 0) {
            unsigned long long t0 = ~0ull, t3 = 0; int n = 0;
            for (int r = 0; r < 2048; ++r) if (tr[r * 4]) { t0 = std::min(t0, tr[r * 4]); t3 = std::max(t3, tr[r * 4 + 3]); ++n; }
            printf("  rows kernel, per-workgroup wall clock of the last launch (10 ns ticks since the first workgroup started; %d workgroups, span %llu):\n", n, t3 - t0);
            printf("   block: start  rows-staged  wave-0-MFMAs-done  end    | phase lengths\n");
            double s[4] = {0, 0, 0, 0}, ph[3] = {0, 0, 0};
            for (int r = 0; r < 2048; ++r) if (tr[r * 4]) {
                const unsigned long long* t = &tr[r * 4];
                if (r % 96 == 0 || t[3] == t3) printf("   %4d: %5llu %5llu %5llu %5llu | %4llu %4llu %4llu%s\n", r, t[0] - t0, t[1] - t0, t[2] - t0, t[3] - t0, t[1] - t[0], t[2] - t[1], t[3] - t[2], t[3] == t3 ? "  <- last" : "");
                for (int k = 0; k < 4; ++k) s[k] += double(t[k] - t0);
                ph[0] += double(t[1] - t[0]); ph[1] += double(t[2] - t[1]); ph[2] += double(t[3] - t[2]);
            }
            printf("   mean: %5.0f %5.0f %5.0f %5.0f | %4.0f %4.0f %4.0f\n", s[0] / n, s[1] / n, s[2] / n, s[3] / n, ph[0] / n, ph[1] / n, ph[2] / n);
        }
    }
    return 0;
}

static int cmd_rvq(int argc, char** argv) {
    const int rows = argc > 2 ? atoi(argv[2]) : 256, iters = argc > 3 ? atoi(argv[3]) : 200;
    const int n_q = 8, dim = 64, size = 1024;
    std::vector<float> he((size_t)n_q * dim * size), hn((size_t)n_q * size);
    std::mt19937 g(7);
    std::normal_distribution<float> nd(0.f, 1.f);
    for (int q = 0; q < n_q; ++q) {
        const float sc = std::pow(0.8f, (float)q);
        for (size_t i = 0; i < (size_t)dim * size; ++i) he[(size_t)q * dim * size + i] = nd(g) * sc;
        for (int c = 0; c < size; ++c) { float s = 0; for (int dd = 0; dd < dim; ++dd) { float v = he[((size_t)q * dim + dd) * size + c]; s += v * v; } hn[(size_t)q * size + c] = s; }
    }
    float *embed, *enorm; CK(hipMalloc(&embed, he.size() * 4)); CK(hipMalloc(&enorm, hn.size() * 4));
    CK(hipMemcpy(embed, he.data(), he.size() * 4, hipMemcpyHostToDevice)); CK(hipMemcpy(enorm, hn.data(), hn.size() * 4, hipMemcpyHostToDevice));
    float* z = dev_random((size_t)rows * dim, 1.f, 11);
    int64_t* idx; CK(hipMalloc(&idx, (size_t)n_q * rows * 8));
    float* zq; CK(hipMalloc(&zq, (size_t)rows * dim * 4));
    hipStream_t st; CK(hipStreamCreate(&st));
    for (int i = 0; i < 5; ++i) AK(adk_rvq_encode(z, embed, enorm, idx, zq, rows, n_q, dim, size, st));
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    CK(hipStreamSynchronize(st));
    CK(hipEventRecord(e0, st));
    for (int i = 0; i < iters; ++i) AK(adk_rvq_encode(z, embed, enorm, idx, zq, rows, n_q, dim, size, st));
    CK(hipEventRecord(e1, st));
    CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    std::vector<int64_t> hi((size_t)n_q * rows); std::vector<float> hq((size_t)rows * dim);
    CK(hipMemcpy(hi.data(), idx, hi.size() * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(hq.data(), zq, hq.size() * 4, hipMemcpyDeviceToHost));
    unsigned long long ck = 1469598103934665603ull; for (int64_t v : hi) { ck ^= (unsigned long long)v; ck *= 1099511628211ull; }
    unsigned long long cq = 1469598103934665603ull; for (float v : hq) { unsigned u; memcpy(&u, &v, 4); cq ^= u; cq *= 1099511628211ull; }
    printf("rvq rows=%d  %8.2f us per launch   idx checksum %016llx  zq checksum %016llx  (variant %s)\n", rows, 1e3 * ms / iters, ck, cq,
           getenv("ADK_RVQ_V1") ? "v1" : getenv("ADK_RVQ_V") ? getenv("ADK_RVQ_V") : "default");
    return 0;
}

int main(int argc, char** argv) {
    if (argc >= 2 && !strcmp(argv[1], "conv")) return cmd_conv(argc, argv);
    if (argc >= 2 && !strcmp(argv[1], "rvq")) return cmd_rvq(argc, argv);
    fprintf(stderr, "usage: kbench conv <shape> <impl> [B] [iters] [ref_impl] | kbench rvq [rows] [iters]\n");
    return 1;
}
