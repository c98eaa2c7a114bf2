// Raw v_mfma_f32_32x32x2_f32 issue-rate probe: NACC independent accumulators per wave, W waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
template <int NACC>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}
template <int NACC>
void run(int blocks_per_cu, int iters) {
    float* out; hipMalloc(&out, 256 * 8 * 256 * 4 * sizeof(float));
    int grid = 256 * blocks_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC><<<grid, 256>>>(out, 10, 1.f, 2.f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    k<NACC><<<grid, 256>>>(out, iters, 1.f, 2.f);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)grid * 4 * iters * 8 * NACC * 4096.0;
    printf("NACC=%d blocks/CU=%d (waves/SIMD=%d): %.3f ms  %.1f TFLOP/s\n", NACC, blocks_per_cu, blocks_per_cu, ms, flops / ms / 1e9);
    hipFree(out);
}
int main() {
    for (int bpc = 1; bpc <= 4; bpc *= 2) { run<1>(bpc, 4000); run<2>(bpc, 2000); run<4>(bpc, 1000); }
    return 0;
}
