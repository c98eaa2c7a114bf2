// "Rows in LDS" causal conv on the fp32 matrix cores -- the time-rich / channel-poor regime
// (stride 1, 32 or 64 input channels per group: encoder blocks 0-1, decoder blocks 2-3, vocoder
// stages 2-3; SURVEY.md section 7 "time-rich / channel-poor").
//
// One workgroup = one (stream, group, time tile): the tile's input rows INCLUDING the causal history
// (the reference's pad_buffer, layers/conv_layer.py:153-156) are staged ONCE into LDS with the input
// activation applied -- every tap then reads them at a shifted row offset, so the state/activation
// rows are fetched from HBM/L2 once instead of once per tap, the ELU is evaluated once per element
// instead of once per tap, and the main loop has no global X traffic and no barrier at all:
//   for tap: for 8-k group: A fragment (pre-packed weights, global/L2 -> VGPR, prefetched one tap
//   ahead) x B fragment (ds_read_b128 at row t + tap*dilation) -> 4 MFMAs per accumulator.
// Waves split the (m-tile, n-tile) pairs of the tile; the split is rotated by the workgroup index so
// the SIMDs of a CU that hosts several workgroups get equal shares.
#include "adk_common.h"
#include <type_traits>

namespace adk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef unsigned u32x4r __attribute__((ext_vector_type(4)));

struct RlArgs {
    int tt;               // time-tile length (output steps per workgroup), multiple of 32 unless one tile covers t_out
    int tiles_per_stream;
    int kgroups;          // 8-k fragments per 32-row m-tile (K padded to a multiple of 64)
    int mt32_per_g;
    int span;             // (taps-1)*dilation history rows in front of a tile
    unsigned w_bytes;
};

__device__ __forceinline__ float4 rl_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4r v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

template <int ACT>
__device__ __forceinline__ float rl_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

template <int C, int ACT, int TAPS>
__global__ __launch_bounds__(256, 3) void conv_rl_kernel(ConvArgs a, RlArgs rl) {
    constexpr int LD = C + 4;                          // LDS row stride (floats): conflict-free b128 reads of consecutive rows
    constexpr int KG = C / 8;                          // 8-k groups per tap
    extern __shared__ __attribute__((aligned(16))) float xs[];     // [(span + tt_pad)][LD]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;

    // workgroup -> (stream b, time tile, group g); groups fastest so neighbours share the input rows in L2
    const int g = blockIdx.x % a.groups;
    const int rest = blockIdx.x / a.groups;
    const int tile = rest % rl.tiles_per_stream;
    const int b = rest / rl.tiles_per_stream;
    const int t0 = tile * rl.tt;
    const int tcur = min(rl.tt, a.t_out - t0);         // output steps of this tile
    const int n_tiles = (tcur + 31) >> 5;

    // ---- stage rows [t0 - span, t0 + 32*n_tiles) once, activation applied ----
    {
        const float* xin = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff + g * a.in_gstride;
        const int rows_valid = rl.span + tcur;          // rows that exist in the ring for this call
        const int rows_all = rl.span + 32 * n_tiles;    // padded so that the last n-tile reads defined memory
        constexpr int C4 = C / 4;
        for (int i = tid; i < rows_all * C4; i += 256) {
            const int rr = i / C4, c4 = i - rr * C4;
            float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
            if (rr < rows_valid) {
                int row = a.in_row0 + t0 + rr;
                row %= a.in_rows;
                v = *reinterpret_cast<const float4*>(xin + (size_t)row * a.in_ch + 4 * c4);
                v.x = rl_act<ACT>(v.x, a.slope); v.y = rl_act<ACT>(v.y, a.slope);
                v.z = rl_act<ACT>(v.z, a.slope); v.w = rl_act<ACT>(v.w, a.slope);
            }
            *reinterpret_cast<float4*>(xs + rr * LD + 4 * c4) = v;
        }
    }
    __syncthreads();

    // ---- split the (m-tile, n-tile) pairs over the 4 waves, rotated by the workgroup index ----
    const int m_tiles = rl.mt32_per_g;
    const int pairs = m_tiles * n_tiles;
    const int slot = (wave + blockIdx.x + (blockIdx.x >> 8)) & 3;   // co-resident workgroups (ids 256 apart) get different rotations
    const int p_begin = (pairs * slot) >> 2, p_end = (pairs * (slot + 1)) >> 2;

    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, rl.w_bytes, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;

    for (int p = p_begin; p < p_end;) {
        const int mt = p / n_tiles, nt0 = p - mt * n_tiles;
        // up to 2 consecutive n-tiles of the same m-tile share each A fragment
        const bool two = (p + 1 < p_end) && (nt0 + 1 < n_tiles);
        const unsigned wbase = (unsigned)((g * m_tiles + mt) * rl.kgroups) * 1024u;
        f32x16 acc0, acc1;
#pragma unroll
        for (int e = 0; e < 16; ++e) { acc0[e] = 0.f; acc1[e] = 0.f; }
        const float* x0 = xs + (nt0 * 32 + l31) * LD + 4 * lh;       // tap 0 row of this lane's column
        const float* x1 = x0 + 32 * LD;
        // The tap loop is fully unrolled (TAPS is a template parameter) with the A fragments ping-ponging
        // between two register sets: in straight-line code hipcc counts its vmcnt waits, so the next tap's
        // weight loads really stay in flight under the current tap's MFMAs (with a rolled loop it waits
        // vmcnt(0) at the loop header and serialises every tap behind an L2 round trip).
        float4 af[2][KG];
#pragma unroll
        for (int q = 0; q < KG; ++q) af[0][q] = rl_load4(rsrc_w, lane16, wbase + (unsigned)q * 1024u);
#pragma unroll
        for (int tap = 0; tap < TAPS; ++tap) {
            if (tap + 1 < TAPS) {
#pragma unroll
                for (int q = 0; q < KG; ++q) af[(tap + 1) & 1][q] = rl_load4(rsrc_w, lane16, wbase + (unsigned)((tap + 1) * KG + q) * 1024u);
            }
            const int roff = tap * a.dilation * LD;
#pragma unroll
            for (int q = 0; q < KG; ++q) {
                const float4 av = af[tap & 1][q];
                const float4 b0 = *reinterpret_cast<const float4*>(x0 + roff + 8 * q);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b0.x, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b0.y, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b0.z, acc0, 0, 0, 0);
                acc0 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b0.w, acc0, 0, 0, 0);
                if (two) {
                    const float4 b1 = *reinterpret_cast<const float4*>(x1 + roff + 8 * q);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, b1.x, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, b1.y, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, b1.z, acc1, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, b1.w, acc1, 0, 0, 0);
                }
            }
        }
        // ---- epilogue: bias, residual, output activation, store (lane = output step, 4 channels per piece) ----
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two) break;
            const f32x16& acc = j ? acc1 : acc0;
            const int t = t0 + (nt0 + j) * 32 + l31;
            if (t >= t0 + tcur) continue;
            const float* resp = nullptr;
            if (a.res) {
                int rrow = a.res_cursor + t;
                if (rrow >= a.res_rows) rrow -= a.res_rows;
                resp = a.res + ((size_t)b * a.res_rows + rrow) * a.res_ch + a.res_choff + g * a.res_gstride;
            }
            int orow = a.out_cursor + t;
            if (orow >= a.out_rows) orow -= a.out_rows;
            float* outp = a.out + ((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff + g * a.cout_g;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = mt * 32 + 8 * qd + 4 * lh;
                if (ml >= a.cout_g) continue;
                float4 v = make_float4(acc[4 * qd], acc[4 * qd + 1], acc[4 * qd + 2], acc[4 * qd + 3]);
                if (a.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(a.bias + g * a.cout_g + ml);
                    v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                }
                if (resp) {
                    const float4 rr = *reinterpret_cast<const float4*>(resp + ml);
                    v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                }
                if (a.act_out != ADK_ACT_NONE) {
                    v.x = act_apply(v.x, a.act_out, 0.f); v.y = act_apply(v.y, a.act_out, 0.f);
                    v.z = act_apply(v.z, a.act_out, 0.f); v.w = act_apply(v.w, a.act_out, 0.f);
                }
                *reinterpret_cast<float4*>(outp + ml) = v;
            }
        }
        p += two ? 2 : 1;
    }
}

// shapes the kernel takes (on top of conv_mfma_supported): stride-1, non-transposed, 32 or 64 channels per group,
// whole 32-row m-tiles, at least one 32-step n-tile worth of work per stream
bool conv_rl_supported(const ConvArgs& a) {
    if (!conv_mfma_supported(a)) return false;
    if (a.stride != 1 || a.up != 1) return false;
    if (a.cin_g != 32 && a.cin_g != 64) return false;
    if (a.taps != 3 && a.taps != 7 && a.taps != 11) return false;    // the tap loop is unrolled at compile time
    if (a.cout_g % 32 != 0) return false;
    if (a.t_out < 24) return false;                    // time-poor: the stream-K kernel batches columns across streams
    return true;
}

// AUTO mode: one workgroup per (stream, group, time tile) only pays when that fills the chip; few streams
// (single-stream latency) go to the stream-K kernel, which spreads one tile's K over all CUs
bool conv_rl_preferred(const ConvArgs& a) {
    if (!conv_rl_supported(a) || a.taps == 1) return false;
    const int ld = a.cin_g + 4;
    int tt = ((54000 / (ld * 4) - (a.taps - 1) * a.dilation) / 32) * 32;
    if (tt < 32) return false;
    const long long tiles = tt >= a.t_out ? 1 : (a.t_out + tt - 1) / tt;
    return (long long)a.batch * tiles * a.groups >= 192;
}

namespace {
template <int C>
int launch_rl(const ConvArgs& a, hipStream_t s) {
    RlArgs rl;
    rl.span = (a.taps - 1) * a.dilation;
    rl.kgroups = (a.ktot + 63) / 64 * 8;
    rl.mt32_per_g = a.cout_g / 32;
    rl.w_bytes = (unsigned)((unsigned long long)a.groups * rl.mt32_per_g * rl.kgroups * 1024ull);
    // time tile: as many steps as fit a third of the CU's 160 KiB of LDS (3 workgroups per CU), whole n-tiles;
    // 54,000 B lets one tile cover a 300-step frame of a 32-channel layer with 50 rows of history
    constexpr int LD = C + 4;
    const int max_rows = 54000 / (LD * 4);
    int tt = ((max_rows - rl.span) / 32) * 32;
    if (tt < 32) return fail(ADK_ERR_SHAPE, "conv: history too long for the rows-in-LDS kernel");
    if (tt >= a.t_out) tt = a.t_out;                   // one tile per stream
    rl.tt = tt;
    rl.tiles_per_stream = (a.t_out + tt - 1) / tt;
    const int tt_pad = (std::min(tt, a.t_out) + 31) / 32 * 32;
    const size_t lds = (size_t)(rl.span + tt_pad) * LD * sizeof(float);
    const long long blocks = (long long)a.batch * rl.tiles_per_stream * a.groups;
    if (blocks > 0x7fffffffLL) return fail(ADK_ERR_SHAPE, "conv: too many workgroups");
    auto go = [&](auto kern) -> int {
        if (lds > 64 * 1024) return fail(ADK_ERR_SHAPE, "conv: rows-in-LDS tile exceeds 64 KiB");
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(256), lds, s, a, rl);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    auto by_taps = [&](auto act) -> int {
        constexpr int ACT = decltype(act)::value;
        if (a.taps == 3) return go(conv_rl_kernel<C, ACT, 3>);
        if (a.taps == 7) return go(conv_rl_kernel<C, ACT, 7>);
        return go(conv_rl_kernel<C, ACT, 11>);
    };
    if (a.act_in == ADK_ACT_ELU) return by_taps(std::integral_constant<int, ADK_ACT_ELU>());
    if (a.act_in == ADK_ACT_LEAKY) return by_taps(std::integral_constant<int, ADK_ACT_LEAKY>());
    if (a.act_in == ADK_ACT_NONE) return by_taps(std::integral_constant<int, ADK_ACT_NONE>());
    return fail(ADK_ERR_ARG, "conv: unsupported input activation for the rows-in-LDS kernel");
}
}  // namespace

int launch_conv_rl(const ConvArgs& a, hipStream_t s) {
    if (a.n_total == 0) return ADK_OK;
    return a.cin_g == 32 ? launch_rl<32>(a, s) : launch_rl<64>(a, s);
}

}  // namespace adk
