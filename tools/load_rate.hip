// Hardware probe: what does ONE CU pull out of L2 (and out of its own L1) per clock, as a function of the load width, the number of
// waves per CU, the loads in flight per wave and of how many CUs do it at the same time?  (tools/, not part of the library.)
//   hipcc --offload-arch=gfx950 -O3 -o load_rate tools/load_rate.hip && ./load_rate
// Every wave walks a window of `span` bytes of one L2-resident buffer `reps` times with U independent loads in flight per lane;
// `share` = 1: all waves of a workgroup read the SAME addresses (a weight stream shared by the waves -- L1 hits for all but the first).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(2); } } while (0)

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));

template <int W, int U>   // W = bytes per lane per load (4, 8, 16)
__global__ __launch_bounds__(1024) void stream_kernel(const unsigned* __restrict__ buf, unsigned* __restrict__ out, unsigned span, int reps, int share, unsigned buf_bytes) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    // window of this wave: workgroups start at different places of the buffer, waves of a workgroup at consecutive windows (or the same)
    const unsigned wg_base = (unsigned)(((unsigned long long)blockIdx.x * 2654435761ull) % (buf_bytes / 4096)) * 4096u;
    const unsigned base = (wg_base + (share ? 0u : (unsigned)wave * span)) % buf_bytes;
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(buf), 0, buf_bytes, 0x00020000);
    unsigned acc = 0;
    const unsigned step = 64u * W;                       // bytes one wave instruction covers
    for (int r = 0; r < reps; ++r) {
        for (unsigned off = 0; off < span; off += step * U) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                unsigned a = base + off + u * step + lane * W;
                if (a >= buf_bytes) a -= buf_bytes;
                if constexpr (W == 16) { const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, a, 0, 0); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
                else if constexpr (W == 8) { const u32x2 v = __builtin_amdgcn_raw_buffer_load_b64(rsrc, a, 0, 0); acc ^= v.x ^ v.y; }
                else { acc ^= __builtin_amdgcn_raw_buffer_load_b32(rsrc, a, 0, 0); }
            }
        }
    }
    if (acc == 0x12345678u) out[blockIdx.x * nw + wave] = acc;   // never true for the data below; keeps the loads alive
}

template <int W, int U>
static void run(const unsigned* buf, unsigned* out, unsigned buf_bytes, int wgs, int waves, unsigned span, int share) {
    const int reps = 8;
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int warm = 0; warm < 2; ++warm) hipLaunchKernelGGL((stream_kernel<W, U>), dim3(wgs), dim3(64 * waves), 0, 0, buf, out, span, reps, share, buf_bytes);
    CK(hipEventRecord(e0));
    const int launches = 5;
    for (int i = 0; i < launches; ++i) hipLaunchKernelGGL((stream_kernel<W, U>), dim3(wgs), dim3(64 * waves), 0, 0, buf, out, span, reps, share, buf_bytes);
    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
    float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
    const double us = ms * 1000.0 / launches;
    const double bytes_wg = (double)span * reps * waves;           // bytes delivered to registers per workgroup
    const int cus = wgs < 256 ? wgs : 256;
    const double per_cu = bytes_wg * wgs / cus / (us * 1e-6) / 1e9;
    printf("W=%2d U=%2d wgs=%4d waves=%2d span=%7u share=%d : %8.1f us  %7.1f GB/s per CU  %6.2f TB/s chip\n", W, U, wgs, waves, span, share, us, per_cu,
           bytes_wg * wgs / (us * 1e-6) / 1e12);
    CK(hipEventDestroy(e0)); CK(hipEventDestroy(e1));
}

int main(int argc, char** argv) {
    // default 2 MiB: every XCD's L2 (4 MiB) holds all of it after the warm-up, like a layer's weights; 16+ MiB: mostly Infinity Cache
    const unsigned buf_bytes = (argc > 1 ? (unsigned)atoi(argv[1]) : 2u) << 20;
    unsigned* buf; unsigned* out;
    CK(hipMalloc(&buf, buf_bytes)); CK(hipMalloc(&out, 1 << 20));
    std::vector<unsigned> h(buf_bytes / 4);
    for (size_t i = 0; i < h.size(); ++i) h[i] = (unsigned)(i * 2654435761u) | 1u;
    CK(hipMemcpy(buf, h.data(), buf_bytes, hipMemcpyHostToDevice));
    const unsigned span = 256u << 10;                     // 256 KiB per wave window (a codebook / a weight block)
    for (int wgs : {1, 64, 256}) {
        for (int share : {0, 1}) {
            run<16, 4>(buf, out, buf_bytes, wgs, 4, span, share);
            run<16, 8>(buf, out, buf_bytes, wgs, 4, span, share);
            run<16, 16>(buf, out, buf_bytes, wgs, 4, span, share);
            run<16, 8>(buf, out, buf_bytes, wgs, 8, span, share);
            run<16, 8>(buf, out, buf_bytes, wgs, 16, span, share);
            run<16, 16>(buf, out, buf_bytes, wgs, 16, span, share);
            run<8, 16>(buf, out, buf_bytes, wgs, 16, span, share);
            run<4, 16>(buf, out, buf_bytes, wgs, 16, span, share);
            run<4, 32>(buf, out, buf_bytes, wgs, 16, span, share);
        }
    }
    // two and three workgroups per CU (4 waves each): the chain kernels' geometry
    for (int wgs : {512, 768}) { run<16, 8>(buf, out, buf_bytes, wgs, 4, span, 0); run<16, 8>(buf, out, buf_bytes, wgs, 4, span, 1); }
    // a small window that fits L1 (16 KiB): the L1 -> register rate
    for (int waves : {4, 16}) { run<16, 8>(buf, out, buf_bytes, 256, waves, 16u << 10, 1); run<4, 16>(buf, out, buf_bytes, 256, waves, 16u << 10, 1); }
    return 0;
}
