// Generic VALU causal conv: one thread per (column n, GEMM row m).  Covers the shapes the MFMA
// implicit-GEMM kernel does not take (Cin = 1 input conv, Cout = 1 output conv, odd channel
// counts) and serves as the in-library cross-check for it (ADK_IMPL_DIRECT).
// Replaces F.conv1d / F.conv_transpose1d as called from layers/conv_layer.py:156,197.
#include "adk_common.h"

namespace adk {

// Lanes run along m (channel axis): the weight rows differ per lane, the input row is shared
// (broadcast) and the store is coalesced along the channel-last output row.
__global__ __launch_bounds__(256) void conv_direct_kernel(ConvArgs a) {
    const int M = a.groups * a.cout_g;
    const long long total = (long long)a.n_total * M;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total;
         gid += (long long)gridDim.x * blockDim.x) {
        const int m = (int)(gid % M);
        const int n = (int)(gid / M);
        const int b = n / a.t_out, t = n - b * a.t_out;
        const int g = m / a.cout_g;
        const float* wrow = a.w + (size_t)m * a.ktot;
        const float* xin = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff + g * a.in_gstride;
        float acc = 0.f;
        for (int j = 0; j < a.taps; ++j) {
            int row = a.in_row0 + t * a.stride + j * a.dilation;
            row %= a.in_rows;
            const float* p = xin + (size_t)row * a.in_ch;
            const float* wj = wrow + j * a.cin_g;
            for (int ci = 0; ci < a.cin_g; ++ci)
                acc = fmaf(wj[ci], act_apply(p[ci], a.act_in, a.slope), acc);
        }
        if (a.bias) acc += a.bias[m];
        if (a.res) {
            int rrow = (a.res_cursor + t) % a.res_rows;
            acc += a.res[((size_t)b * a.res_rows + rrow) * a.res_ch + a.res_choff + g * a.res_gstride + (m - g * a.cout_g)];
        }
        acc = act_apply(acc, a.act_out, 0.f);
        int orow = (a.out_cursor + t * a.up + m / a.cout_real) % a.out_rows;
        a.out[((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff + (m % a.cout_real)] = acc;
    }
}

// Cout_total == 1 (output conv 32 -> 1, K7): lanes run along time; each lane walks taps*cin
// contiguous-per-row inputs with float4 loads, weights come from LDS (broadcast).
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvArgs a) {
    extern __shared__ float wsh[];
    for (int i = threadIdx.x; i < a.ktot; i += blockDim.x) wsh[i] = a.w[i];
    __syncthreads();
    const int n = blockIdx.x * blockDim.x + threadIdx.x;
    if (n >= a.n_total) return;
    const int b = n / a.t_out, t = n - b * a.t_out;
    const float* xin = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff;
    float acc = 0.f;
    for (int j = 0; j < a.taps; ++j) {
        int row = a.in_row0 + t * a.stride + j * a.dilation;
        row %= a.in_rows;
        const float4* p = reinterpret_cast<const float4*>(xin + (size_t)row * a.in_ch);
        const float* wj = wsh + j * a.cin_g;
        for (int c4 = 0; c4 < a.cin_g / 4; ++c4) {
            float4 v = p[c4];
            acc = fmaf(wj[4 * c4 + 0], act_apply(v.x, a.act_in, a.slope), acc);
            acc = fmaf(wj[4 * c4 + 1], act_apply(v.y, a.act_in, a.slope), acc);
            acc = fmaf(wj[4 * c4 + 2], act_apply(v.z, a.act_in, a.slope), acc);
            acc = fmaf(wj[4 * c4 + 3], act_apply(v.w, a.act_in, a.slope), acc);
        }
    }
    if (a.bias) acc += a.bias[0];
    if (a.res) {
        int rrow = (a.res_cursor + t) % a.res_rows;
        acc += a.res[((size_t)b * a.res_rows + rrow) * a.res_ch + a.res_choff];
    }
    acc = act_apply(acc, a.act_out, 0.f);
    int orow = (a.out_cursor + t) % a.out_rows;
    a.out[((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff] = acc;
}

int launch_conv_direct(const ConvArgs& a, hipStream_t s) {
    const int M = a.groups * a.cout_g;
    if (a.n_total == 0) return ADK_OK;
    const bool aligned = (a.in_ch % 4 == 0) && (a.in_choff % 4 == 0) && (a.cin_g % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(a.in) & 15) == 0);
    if (M == 1 && a.up == 1 && aligned && a.ktot * sizeof(float) <= 48 * 1024) {
        const int blocks = (a.n_total + 255) / 256;
        hipLaunchKernelGGL(conv_cout1_kernel, dim3(blocks), dim3(256), a.ktot * sizeof(float), s, a);
    } else {
        const long long total = (long long)a.n_total * M;
        long long blocks = (total + 255) / 256;
        if (blocks > 65536 * 4) blocks = 65536 * 4;
        hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

}  // namespace adk
