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

// Cin_total == 1 (encoder input conv 1 -> 32, K7): pure streaming -- 4 B in, 128 B out per step.
// Cout/4 threads per time step, 4 output channels each: the weights of a thread's 4 channels sit in
// registers for the whole launch, the 7 input samples are broadcast loads, the store is one coalesced float4.
template <int TAPS>
__global__ __launch_bounds__(256) void conv_cin1_kernel(ConvArgs a) {
    const int M = a.cout_g;                                 // groups == 1
    const int q = M / 4;                                    // float4 pieces per output row; q divides 256
    const int c4 = threadIdx.x % q;                         // fixed per thread: its 4 channels' weights stay in registers
    float w[4][TAPS];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < TAPS; ++j) w[k][j] = a.w[(size_t)(4 * c4 + k) * TAPS + j];
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + 4 * c4);
    const int per_block = 256 / q;                          // time steps per workgroup pass
    for (long long n0 = (long long)blockIdx.x * per_block; n0 < a.n_total; n0 += (long long)gridDim.x * per_block) {
        const int n = (int)n0 + threadIdx.x / q;
        if (n >= a.n_total) continue;
        const int b = n / a.t_out, t = n - b * a.t_out;
        const float* xin = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff;
        float4 acc = bias;
        int row = a.in_row0 + t * a.stride;
        if (row >= a.in_rows) row -= a.in_rows;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            const float x = act_apply(xin[(size_t)row * a.in_ch], a.act_in, a.slope);
            acc.x = fmaf(w[0][j], x, acc.x); acc.y = fmaf(w[1][j], x, acc.y);
            acc.z = fmaf(w[2][j], x, acc.z); acc.w = fmaf(w[3][j], x, acc.w);
            row += a.dilation;
            if (row >= a.in_rows) row -= a.in_rows;
        }
        int orow = a.out_cursor + t;
        if (orow >= a.out_rows) orow -= a.out_rows;
        *reinterpret_cast<float4*>(a.out + ((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff + 4 * c4) = acc;
    }
}

// The same conv when the ring write of the caller's samples stands right in front of it (the first two ops of the encoder program:
// audio -> ring, ring -> 32 channels): ONE launch.  New samples are read from the caller's buffer src[b][t] (the ring would hand back the very
// same values), the TAPS - 1 samples in front of the step from the ring -- what earlier steps left there --, and the lanes of channel piece 0
// copy the step's samples into the ring for the steps to come (and for a replay of this one: ADK_STEP_REPLAY runs the plain kernel on the
// ring).  Same fma order as conv_cin1_kernel: bit-identical.
template <int TAPS>
__global__ __launch_bounds__(256) void conv_cin1w_kernel(ConvArgs a, const float* __restrict__ src, float* __restrict__ ring) {
    const int M = a.cout_g;
    const int q = M / 4;
    const int c4 = threadIdx.x % q;
    float w[4][TAPS];
#pragma unroll
    for (int k = 0; k < 4; ++k)
#pragma unroll
        for (int j = 0; j < TAPS; ++j) w[k][j] = a.w[(size_t)(4 * c4 + k) * TAPS + j];
    float4 bias = make_float4(0.f, 0.f, 0.f, 0.f);
    if (a.bias) bias = *reinterpret_cast<const float4*>(a.bias + 4 * c4);
    const int per_block = 256 / q;
    for (long long n0 = (long long)blockIdx.x * per_block; n0 < a.n_total; n0 += (long long)gridDim.x * per_block) {
        const int n = (int)n0 + threadIdx.x / q;
        if (n >= a.n_total) continue;
        const int b = n / a.t_out, t = n - b * a.t_out;
        const float* xs = src + (size_t)b * a.t_out;
        float* xr = ring + (size_t)b * a.in_rows * a.in_ch + a.in_choff;
        float4 acc = bias;
        int row = a.in_row0 + t;                             // ring row of sample t - (TAPS - 1)
        if (row >= a.in_rows) row -= a.in_rows;
#pragma unroll
        for (int j = 0; j < TAPS; ++j) {
            const int sidx = t + j - (TAPS - 1);
            const float x = act_apply(sidx >= 0 ? xs[sidx] : xr[(size_t)row * a.in_ch], a.act_in, a.slope);
            acc.x = fmaf(w[0][j], x, acc.x); acc.y = fmaf(w[1][j], x, acc.y);
            acc.z = fmaf(w[2][j], x, acc.z); acc.w = fmaf(w[3][j], x, acc.w);
            if (j == TAPS - 1 && c4 == 0) xr[(size_t)row * a.in_ch] = xs[t];      // (row is the ring row of sample t here)
            row += 1;
            if (row >= a.in_rows) row -= a.in_rows;
        }
        int orow = a.out_cursor + t;
        if (orow >= a.out_rows) orow -= a.out_rows;
        *reinterpret_cast<float4*>(a.out + ((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff + 4 * c4) = acc;
    }
}

// Cout_total == 1 (output conv 32 -> 1, K7 + tanh): one workgroup = 256 consecutive steps of one stream.
// The (256 + hist) input rows are staged once into LDS (coalesced float4 loads, activation applied once,
// row stride Cin+1 so that lanes walking consecutive rows hit distinct banks), then each lane forms its
// taps*Cin dot product from LDS with the weights broadcast from LDS.
__global__ __launch_bounds__(256) void conv_cout1_kernel(ConvArgs a) {
    extern __shared__ float sh[];
    const int C = a.cin_g, ld = C + 1;
    const int span = (a.taps - 1) * a.dilation;            // == hist for stride 1
    float* wsh = sh;                                        // [ktot]
    float* xs = sh + a.ktot;                                // [(256 + span)][C + 1]
    const int tiles_per_stream = (a.t_out + 255) / 256;
    const int b = blockIdx.x / tiles_per_stream;
    const int t0 = (blockIdx.x - b * tiles_per_stream) * 256;
    const int nt = min(256, a.t_out - t0);
    for (int i = threadIdx.x; i < a.ktot; i += 256) wsh[i] = a.w[i];
    const float* xin = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff;
    const int rows = nt + span, c4n = C / 4;
    for (int i = threadIdx.x; i < rows * c4n; i += 256) {
        const int rr = i / c4n, c4 = i - rr * c4n;
        int row = a.in_row0 + t0 + rr;
        row %= a.in_rows;
        const float4 v = *reinterpret_cast<const float4*>(xin + (size_t)row * a.in_ch + 4 * c4);
        float* d = xs + rr * ld + 4 * c4;
        d[0] = act_apply(v.x, a.act_in, a.slope); d[1] = act_apply(v.y, a.act_in, a.slope);
        d[2] = act_apply(v.z, a.act_in, a.slope); d[3] = act_apply(v.w, a.act_in, a.slope);
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= nt) return;
    float acc = 0.f;
    for (int j = 0; j < a.taps; ++j) {
        const float* xr = xs + (t + j * a.dilation) * ld;
        const float* wj = wsh + j * C;
        for (int c = 0; c < C; ++c) acc = fmaf(wj[c], xr[c], acc);
    }
    if (a.bias) acc += a.bias[0];
    if (a.res) {
        int rrow = (a.res_cursor + t0 + t) % a.res_rows;
        acc += a.res[((size_t)b * a.res_rows + rrow) * a.res_ch + a.res_choff];
    }
    acc = act_apply(acc, a.act_out, 0.f);
    int orow = (a.out_cursor + t0 + t) % a.out_rows;
    a.out[((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff] = acc;
}

int launch_conv_direct(const ConvArgs& a, hipStream_t s) {
    const int M = a.groups * a.cout_g;
    if (a.n_total == 0) return ADK_OK;
    const bool aligned = (a.in_ch % 4 == 0) && (a.in_choff % 4 == 0) && (a.cin_g % 4 == 0) &&
                         ((reinterpret_cast<uintptr_t>(a.in) & 15) == 0);
    const size_t cout1_lds = ((size_t)a.ktot + (size_t)(256 + (a.taps - 1) * a.dilation) * (a.cin_g + 1)) * sizeof(float);
    if (M == 1 && a.groups == 1 && a.up == 1 && a.stride == 1 && aligned && cout1_lds <= 64 * 1024) {
        const int blocks = a.batch * ((a.t_out + 255) / 256);
        hipLaunchKernelGGL(conv_cout1_kernel, dim3(blocks), dim3(256), cout1_lds, s, a);
    } else if (a.cin_g == 1 && a.groups == 1 && a.up == 1 && a.taps == 7 && M % 4 == 0 && 256 % (M / 4) == 0 && a.out_ch % 4 == 0 &&
               a.out_choff % 4 == 0 && !a.res && a.act_out == ADK_ACT_NONE && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
               (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0)) {
        const int per_block = 256 / (M / 4);
        long long blocks = ((long long)a.n_total + per_block - 1) / per_block;
        if (blocks > 4096) blocks = 4096;
        hipLaunchKernelGGL(conv_cin1_kernel<7>, dim3((unsigned)blocks), dim3(256), 0, s, a);
    } else {
        const long long total = (long long)a.n_total * M;
        long long blocks = (total + 255) / 256;
        if (blocks > 65536 * 4) blocks = 65536 * 4;
        hipLaunchKernelGGL(conv_direct_kernel, dim3((unsigned)blocks), dim3(256), 0, s, a);
    }
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

// the Cin = 1, K7 conv with the ring write of its input in the same launch (conv_cin1w_kernel): a's input view is the ring AFTER the write
// would have happened (in_row0 = write cursor - 6); src = the caller's rows (batch x t_out samples, one channel)
bool conv_cin1_write_ok(const ConvArgs& a) {
    const int M = a.groups * a.cout_g;
    return a.cin_g == 1 && a.groups == 1 && a.up == 1 && a.taps == 7 && a.stride == 1 && a.dilation == 1 && a.w && M % 4 == 0 && 256 % (M / 4) == 0 &&
           a.out_ch % 4 == 0 && a.out_choff % 4 == 0 && !a.res && a.act_out == ADK_ACT_NONE && (reinterpret_cast<uintptr_t>(a.out) & 15) == 0 &&
           (!a.bias || (reinterpret_cast<uintptr_t>(a.bias) & 15) == 0) && a.in_rows >= a.t_out + 6;
}
int launch_conv_cin1_write(const ConvArgs& a, const float* src, hipStream_t s) {
    if (!conv_cin1_write_ok(a) || !src) return ADK_ERR_STATE;
    if (a.n_total == 0) return ADK_OK;
    const int per_block = 256 / (a.cout_g / 4);
    long long blocks = ((long long)a.n_total + per_block - 1) / per_block;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(conv_cin1w_kernel<7>, dim3((unsigned)blocks), dim3(256), 0, s, a, src, const_cast<float*>(a.in));
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

}  // namespace adk
