// Streaming kernel for the LAST up-sampling stage: fused input activation -> ConvTranspose1d(64 -> Cout, K = 2s, stride s) + bias
// in polyphase form (2 taps, s*Cout <= 96 GEMM rows), split-f16 operands.  This is the north-star's named kernel:
//   c = self.upsamples[i].inference(self.activation_upsamples(c))      models/vocoder/HiFiGAN.py:285-289 (i = 3: 64 -> 32, s = 3)
//   CausalConvTranspose1d.inference                                     layers/conv_layer.py:194-197
// (the symmetric decoder's last DecoderBlock uses the same layer without the activation, modules/decoder.py:70-81).
//
// Per stream and frame it reads 101 rows x 256 B and writes 300 rows x 128 B (64 KB) for 2.46 MFLOP: a pure streamer.  The
// general rows-in-LDS kernel (conv_rl16.hip) spends most of its 11 us on this layer in fixed costs -- rows staged through LDS
// behind a barrier, 16 KB of weight fragments re-read from L2 per work item, two dispatch rounds -- so this layer gets its own
// kernel built around ONE round trip to memory:
//   * one workgroup = one stream (x 128 output steps), wave w = time tile w (32 steps) x ALL m-tiles (phases);
//   * the B operand (activations) never touches LDS: every lane loads its MFMA fragment -- 8 consecutive channels of the row
//     of ITS time step, for both taps -- straight from the ring into registers (16 x 16-byte loads in flight per lane), applies
//     the activation and splits into f16 hi / lo there;
//   * the 48 KB of split weights are fetched ONCE per workgroup, in the same round trip, parked in LDS (lane-linear copy of the
//     adk_pack_weights_split16 layout, conflict-free b128 reads) and shared by the four waves; the single barrier is passed
//     while the activation loads are still in flight;
//   * outputs leave as 16-byte stores, the 3 phases of a time step fill one contiguous 384-byte span.
#include "adk_common.h"
#include <type_traits>

namespace adk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8u __attribute__((ext_vector_type(8)));

namespace {
constexpr float kUpLoScale = 2048.f, kUpLoInv = 1.f / 2048.f;
constexpr int UP_CIN = 64, UP_KSTEPS = 2 * UP_CIN / 16;          // 2 taps x 64 channels = 128 k = 8 chunks of 16

template <int ACT>
__device__ __forceinline__ float up_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

struct UpArgs { int chunks_per_stream; int m_tiles; float inv_cout_real; int* err; };
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

template <int ACT, int MT>
__global__ __launch_bounds__(256, 2) void conv_up16_kernel(ConvArgs a, UpArgs u) {
    extern __shared__ __attribute__((aligned(16))) unsigned char wlds[];      // [MT][8 chunks][hi | lo][64 lanes][16 B]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / u.chunks_per_stream;
    const int t0 = (blockIdx.x - b * u.chunks_per_stream) * 128 + wave * 32;   // first output step of this wave
    const int t = t0 + l31;
    const bool live = t0 < a.t_out;                         // wave-uniform
    const bool valid = t < a.t_out;

    // ---- bias (first: the oldest load), then the activations: this lane's B fragments for both taps, straight from the ring ----
    // GEMM column n = time step t; k = (tap j, channel ci), chunk s = 4*j + ci/16; the lane holds k = 16*s + 8*lh + 0..7,
    // i.e. channels 16*(s%4) + 8*lh + 0..7 of ring row (in_row0 + t + j)  (tap 0 = the older row x[t-1])
    constexpr int WBYTES = MT * UP_KSTEPS * 2048;
    float* blds = reinterpret_cast<float*>(wlds + WBYTES);                     // [32 * MT] bias (zeros when there is none)
    float bias_v = 0.f;
    if (a.bias && tid < 32 * MT) bias_v = a.bias[tid];
    float4 xr[UP_KSTEPS][2];
    {   // unconditional (a branch here makes hipcc wait for the loads at the join, before the DMA below is issued)
        const int tt = valid ? t : a.t_out - 1;             // padded columns read a valid row; their results are not stored
        const float* xin = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff + 8 * lh;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            int row = a.in_row0 + tt + j;
            if (row >= a.in_rows) row -= a.in_rows;
            const float4* p = reinterpret_cast<const float4*>(xin + (size_t)row * a.in_ch);
#pragma unroll
            for (int c = 0; c < 4; ++c) { xr[4 * j + c][0] = p[4 * c]; xr[4 * j + c][1] = p[4 * c + 1]; }
        }
    }
    // ---- weights: one lane-linear copy global -> LDS per workgroup (LDS-DMA, no registers), issued in the same round trip as
    // the loads above.  (Register staging: hipcc sinks every weight load to its ds_write and serialises twelve round trips;
    // LDS-DMA issued BEFORE ordinary loads: it drains the queue at the first ordinary load that follows.) ----
    constexpr int WPT = WBYTES / 16 / 256;                  // 16-byte pieces per thread (MT = 3: 12)
    {
        const unsigned char* gsrc = reinterpret_cast<const unsigned char*>(a.wfrag) + (size_t)tid * 16;
        unsigned char* ldst = wlds + wave * 1024;           // wave-uniform base; the lane offset is implicit
#pragma unroll
        for (int i = 0; i < WPT; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(gsrc + 4096 * i), (lptr_t)(ldst + 4096 * i), 16, 0, 0);
    }
    __builtin_amdgcn_sched_barrier(0);                      // nothing below (the conversion of xr) may move above the DMA issue
    if (tid < 32 * MT) blds[tid] = bias_v;
    // every wave's LDS-DMA slice is read by the OTHER waves: drain this wave's VM-counted DMA explicitly before the barrier that
    // publishes it (hipcc 7.2 happens to emit the wait for the workgroup fence; the fence does not formally promise it)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                        // the weights of all waves are in LDS (and the loads above have landed: one round trip for all)
    if (!live) return;

    f16x8u bh[UP_KSTEPS], bl[UP_KSTEPS];
    auto convert = [&](int s) {
        const float x[8] = {xr[s][0].x, xr[s][0].y, xr[s][0].z, xr[s][0].w, xr[s][1].x, xr[s][1].y, xr[s][1].z, xr[s][1].w};
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const float v = up_act<ACT>(x[e], a.slope);
            const _Float16 h = (_Float16)v;
            bh[s][e] = h;
            bl[s][e] = (_Float16)((v - (float)h) * kUpLoScale);
        }
    };

    float* outb = a.out + (size_t)b * a.out_rows * a.out_ch + a.out_choff;
    int orow0 = a.out_cursor + t * a.up;
    orow0 %= a.out_rows;
    bool bad = false;
    convert(0);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
        f32x16 am, ac;
#pragma unroll
        for (int e = 0; e < 16; ++e) { am[e] = 0.f; ac[e] = 0.f; }
        const unsigned char* wp = wlds + (size_t)mt * UP_KSTEPS * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < UP_KSTEPS; ++s) {
            const f16x8u Ah = *reinterpret_cast<const f16x8u*>(wp + s * 2048);
            const f16x8u Al = *reinterpret_cast<const f16x8u*>(wp + s * 2048 + 1024);
            am = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh[s], am, 0, 0, 0);
            if (mt == 0 && s + 1 < UP_KSTEPS) convert(s + 1);       // the split of the next chunk issues under this chunk's MFMAs
            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl[s], ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh[s], ac, 0, 0, 0);
        }
        if (!valid) continue;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int ml = mt * 32 + 8 * qd + 4 * lh;       // GEMM row = phase * cout_real + co   (cout_g == 32 * MT)
            float4 v = make_float4(fmaf(ac[4 * qd], kUpLoInv, am[4 * qd]), fmaf(ac[4 * qd + 1], kUpLoInv, am[4 * qd + 1]),
                                   fmaf(ac[4 * qd + 2], kUpLoInv, am[4 * qd + 2]), fmaf(ac[4 * qd + 3], kUpLoInv, am[4 * qd + 3]));
            bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
            const float4 bb = *reinterpret_cast<const float4*>(blds + ml);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            const int ph = (int)(((float)ml + 0.5f) * u.inv_cout_real);        // ml / cout_real, exact for these sizes
            int r2 = orow0 + ph;
            if (r2 >= a.out_rows) r2 -= a.out_rows;
            *reinterpret_cast<float4*>(outb + (size_t)r2 * a.out_ch + (ml - ph * a.cout_real)) = v;
        }
    }
    if (bad) atomicOr(u.err, 8);
}
}  // namespace

// 2-tap polyphase transposed conv, 64 input channels, up to 96 GEMM rows (the 64 -> 32, stride 3 layer), split16 weights
bool conv_up16_supported(const ConvArgs& a) {
    if (!a.wfrag || a.up <= 1 || a.taps != 2 || a.stride != 1 || a.dilation != 1 || a.groups != 1 || a.res) return false;
    if (a.act_out != ADK_ACT_NONE) return false;
    if (a.cin_g != UP_CIN || a.cout_g % 32 != 0 || a.cout_g > 96 || a.cout_real % 4 != 0) return false;
    if ((a.in_ch % 4) || (a.in_choff % 4) || (a.out_ch % 4) || (a.out_choff % 4)) return false;
    if ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.wfrag)) & 15) return false;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return false;
    return true;
}

int launch_conv_up16(const ConvArgs& a, hipStream_t s) {
    if (a.n_total == 0) return ADK_OK;
    UpArgs u;
    u.chunks_per_stream = (a.t_out + 127) / 128;
    u.m_tiles = a.cout_g / 32;
    u.inv_cout_real = 1.0f / (float)a.cout_real;
    u.err = conv_err_word(a);
    const long long blocks = (long long)a.batch * u.chunks_per_stream;
    if (blocks > 0x7fffffffLL) return fail(ADK_ERR_SHAPE, "conv: too many workgroups");
    const size_t lds = (size_t)u.m_tiles * UP_KSTEPS * 2048 + (size_t)u.m_tiles * 32 * sizeof(float);
    auto go = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a, u);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    auto by_mt = [&](auto act) -> int {
        constexpr int ACT = decltype(act)::value;
        if (u.m_tiles == 1) return go(conv_up16_kernel<ACT, 1>);
        if (u.m_tiles == 2) return go(conv_up16_kernel<ACT, 2>);
        return go(conv_up16_kernel<ACT, 3>);
    };
    if (a.act_in == ADK_ACT_ELU) return by_mt(std::integral_constant<int, ADK_ACT_ELU>());
    if (a.act_in == ADK_ACT_LEAKY) return by_mt(std::integral_constant<int, ADK_ACT_LEAKY>());
    if (a.act_in == ADK_ACT_NONE) return by_mt(std::integral_constant<int, ADK_ACT_NONE>());
    return fail(ADK_ERR_ARG, "conv: unsupported input activation for the up-sampling kernel");
}

}  // namespace adk
