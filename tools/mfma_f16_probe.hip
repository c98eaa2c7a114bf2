// v_mfma_f32_32x32x16_f16 probes for the split-precision (f16 hi + scaled f16 lo, 3 products) conv idea:
//  (1) are subnormal f16 inputs honoured?   (2) accuracy of the 3-product scheme vs fp64   (3) issue rate.
// hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_probe.hip -o /tmp/f16probe && /tmp/f16probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <vector>
#include <random>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__global__ void denorm_kernel(float* out) {
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)9.5367431640625e-07f /* 2^-20: f16 subnormal */; b[i] = (_Float16)1.0f; }
    f32x16 c;
    for (int e = 0; e < 16; ++e) c[e] = 0.f;
    c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) out[0] = c[0];
}

// C[32x32] = A[32xK] * B[Kx32] with the 3-product split; K multiple of 16.  One wave.
__global__ void split_kernel(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kb = l >> 5;
    f32x16 acc0, acc1;
    for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
    for (int k0 = 0; k0 < K; k0 += 16) {
        f16x8 ah, al, bh, bl;
        for (int j = 0; j < 8; ++j) {
            const int k = k0 + 8 * kb + j;
            const float a = A[i * K + k], b = B[k * 32 + i];
            const _Float16 h = (_Float16)a; ah[j] = h; al[j] = (_Float16)((a - (float)h) * 2048.f);
            const _Float16 g = (_Float16)b; bh[j] = g; bl[j] = (_Float16)((b - (float)g) * 2048.f);
        }
        acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
        acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
    }
    for (int r = 0; r < 16; ++r) {
        const int row = (r & 3) + 8 * (r >> 2) + 4 * kb;
        C[row * 32 + i] = acc0[r] + acc1[r] * (1.f / 2048.f);
    }
}

__global__ void f32_kernel(const float* A, const float* B, float* C, int K) {
    const int l = threadIdx.x, i = l & 31, kb = l >> 5;
    f32x16 acc;
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(A[i * K + k0 + kb], B[(k0 + kb) * 32 + i], acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) C[((r & 3) + 8 * (r >> 2) + 4 * kb) * 32 + i] = acc[r];
}

template <int NACC>
__global__ __launch_bounds__(256) void rate_kernel(float* out, int iters) {
    f32x16 acc[NACC];
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(1.0f + threadIdx.x * 0.001f); b[i] = (_Float16)0.5f; }
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u)
#pragma unroll
            for (int j = 0; j < NACC; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[j], 0, 0, 0);
    }
    float s = 0;
    for (int j = 0; j < NACC; ++j) for (int e = 0; e < 16; ++e) s += acc[j][e];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

int main() {
    float* d; hipMalloc(&d, 1 << 22);
    denorm_kernel<<<1, 64>>>(d);
    float h; hipMemcpy(&h, d, 4, hipMemcpyDeviceToHost);
    printf("subnormal f16 input: got %.9g, expect %.9g (16 * 2^-20) -> %s\n", h, 16 * 9.5367431640625e-07, h > 0 ? "honoured" : "FLUSHED");
    for (int K : {352, 1408, 2816}) {
        std::vector<float> A(32 * K), B(K * 32), C(1024), C32(1024);
        std::mt19937 g(K); std::normal_distribution<float> nd(0.f, 1.f);
        for (auto& v : A) v = 0.05f * nd(g);
        for (auto& v : B) { v = nd(g); if (v < 0) v *= 0.1f; }          // LeakyReLU-like activations
        float *dA, *dB, *dC; hipMalloc(&dA, A.size() * 4); hipMalloc(&dB, B.size() * 4); hipMalloc(&dC, 4096);
        hipMemcpy(dA, A.data(), A.size() * 4, hipMemcpyHostToDevice); hipMemcpy(dB, B.data(), B.size() * 4, hipMemcpyHostToDevice);
        split_kernel<<<1, 64>>>(dA, dB, dC, K); hipMemcpy(C.data(), dC, 4096, hipMemcpyDeviceToHost);
        f32_kernel<<<1, 64>>>(dA, dB, dC, K); hipMemcpy(C32.data(), dC, 4096, hipMemcpyDeviceToHost);
        double es = 0, e32 = 0, ref_abs = 0;
        for (int i = 0; i < 32; ++i) for (int j = 0; j < 32; ++j) {
            double r = 0, ra = 0;
            for (int k = 0; k < K; ++k) { r += (double)A[i * K + k] * B[k * 32 + j]; ra += fabs((double)A[i * K + k] * B[k * 32 + j]); }
            es = fmax(es, fabs(C[i * 32 + j] - r) / ra); e32 = fmax(e32, fabs(C32[i * 32 + j] - r) / ra); ref_abs = fmax(ref_abs, ra);
        }
        printf("K=%4d  max |err| / sum|a b|:  f16x3 %.3e   f32 mfma %.3e\n", K, es, e32);
        hipFree(dA); hipFree(dB); hipFree(dC);
    }
    for (int bpc = 1; bpc <= 4; bpc *= 2) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int grid = 256 * bpc, iters = 4000;
        rate_kernel<4><<<grid, 256>>>(d, 10); hipDeviceSynchronize();
        hipEventRecord(e0); rate_kernel<4><<<grid, 256>>>(d, iters); hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("f16 32x32x16, 4 acc, %d waves/SIMD: %.3f ms  %.0f TFLOP/s\n", bpc, ms, (double)grid * 4 * iters * 8 * 4 * 32768.0 / ms / 1e9);
    }
    return 0;
}
