// MFMA + prefetching global loads probe (mimics the conv main loop without LDS/barriers).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int MODE>   // 0: MFMA only, 1: + 6 prefetch loads / 32 MFMA (consumed next iter), 2: + LDS write/read + barrier
__global__ __launch_bounds__(256, 2) void k(const float4* __restrict__ src, float* out, int iters, int span) {
    __shared__ float4 lds[2][1152];
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    const int tid = threadIdx.x;
    const float4* p = src + (blockIdx.x * 977 % span) * 256 + tid;
    float4 a_cur[4], a_nxt[4], b_nxt[2];
    for (int q = 0; q < 4; ++q) a_cur[q] = p[q * 64];
    int cur = 0;
    for (int i = 0; i < iters; ++i) {
        if (MODE >= 1) {
            const float4* pn = src + ((blockIdx.x * 977 + (i + 1) * 131) % span) * 256 + tid;
            for (int q = 0; q < 4; ++q) a_nxt[q] = pn[q * 64];
            b_nxt[0] = pn[300]; b_nxt[1] = pn[700];
        }
        float4 bv[4];
        if (MODE >= 2) { for (int q = 0; q < 4; ++q) bv[q] = lds[cur][(tid & 63) * 9 + q * 2]; }
        else { for (int q = 0; q < 4; ++q) bv[q] = a_cur[(q + 1) & 3]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].x, bv[q].x, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].x, bv[q].y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].y, bv[q].y, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].y, bv[q].z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].z, bv[q].z, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].z, bv[q].w, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].w, bv[q].w, acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(a_cur[q].w, bv[q].x, acc1, 0, 0, 0);
        }
        if (MODE >= 2) {
            lds[cur ^ 1][tid * 2] = b_nxt[0]; lds[cur ^ 1][tid * 2 + 1] = b_nxt[1];
            __syncthreads();
            cur ^= 1;
        }
        if (MODE >= 1) { for (int q = 0; q < 4; ++q) a_cur[q] = a_nxt[q]; if (MODE == 1) { a_cur[0].x += b_nxt[0].x + b_nxt[1].y; } }
    }
    float s = 0;
    for (int e = 0; e < 16; ++e) s += acc0[e] + acc1[e];
    out[blockIdx.x * 256 + tid] = s;
}
template <int MODE>
void run(int bpc, int iters, int span_mb) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * sizeof(float));
    float4* src; size_t n = (size_t)span_mb * 1024 * 1024 / 16 + 4096; hipMalloc(&src, n * 16); hipMemset(src, 0, n * 16);
    int span = (int)((size_t)span_mb * 1024 * 1024 / 16 / 256) - 8;
    int grid = 256 * bpc;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256>>>(src, out, 10, span); hipDeviceSynchronize();
    hipEventRecord(e0);
    k<MODE><<<grid, 256>>>(src, out, iters, span);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("MODE=%d blocks/CU=%d span=%dMB: %.3f ms  %.1f TFLOP/s  (%.0f cycles/iter @2.4GHz)\n", MODE, bpc, span_mb, ms, flops / ms / 1e9,
           ms * 1e-3 * 2.4e9 / iters / bpc);
    hipFree(out); hipFree(src);
}
int main() {
    for (int bpc = 1; bpc <= 2; ++bpc) {
        run<0>(bpc, 1000, 2); run<1>(bpc, 1000, 2); run<1>(bpc, 1000, 64); run<1>(bpc, 1000, 1024);
        run<2>(bpc, 1000, 2); run<2>(bpc, 1000, 64);
    }
    return 0;
}
