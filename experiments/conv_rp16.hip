// Pipelined "rows in LDS" causal conv, split-f16 operands: persistent workgroups that stage the rows of their NEXT work item
// between the MFMA steps of the current one.
//
// Same arithmetic, operands, LDS row layout and per-output MFMA order as conv_rl16_kernel (conv_rl16.hip: one workgroup = one
// (stream, group, time tile), rows + history staged once, then one pass of MFMAs) -- results are bit-identical to it -- and
// the same place on the path: F.conv1d of CausalConv1d.inference (layers/conv_layer.py:153-156) fused with the input
// activation, bias and residual add, for the grouped K11 convs of vocoder stages 2-3 (HiFiGANResidualBlock.inference,
// modules/residual_block.py:99-105: 12 launches of a 256-stream step, a quarter of its kernel time).
//
// Why.  Per-workgroup wall clocks of conv_rl16 on those layers (profiles/r2_rl16_workgroup_clocks.log): every workgroup stages
// for 4-5 us, multiplies for 14-20 us and stores for 3-7 us, and because a launch is ONE round of three workgroups per CU that
// all start together, the co-resident ones stage, multiply and store at the same time -- 13 us of matrix-core work in a 32 us
// span.  Here a CU holds one workgroup with TWO row buffers; it walks over its items (blockIdx, blockIdx + G, ...), and while
// its waves run the MFMAs of item i every thread converts one staging piece of item i + 1 every few k-steps (loads issued one
// piece ahead), so the vector-memory, VALU and matrix-core work of consecutive items overlap inside each wave.
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

namespace adk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8p __attribute__((ext_vector_type(8)));
typedef unsigned u32x4p __attribute__((ext_vector_type(4)));

#ifndef ADK_RP16_DBG
#define ADK_RP16_DBG 0      // 1: wall-clock stamps of the items of workgroups 0 and 100, wave 0 (tools/kbench prints them)
#endif
#if ADK_RP16_DBG & 1
__device__ unsigned long long g_rp_trace[2 * 16 * 8];
extern "C" int adk_debug_rp_trace(unsigned long long* out, int n) {
    if (n > 2 * 16 * 8) n = 2 * 16 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rp_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#define RP_STAMP(slot) do { if (tr_on && it < 16 && lane == 0) tr_base[it * 8 + (slot)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define RP_STAMP(slot) do { } while (0)
#endif

namespace {
constexpr float kRpLoScale = 2048.f, kRpLoInv = 1.f / 2048.f;

struct RpArgs {
    int tt;               // time-tile length (output steps per item)
    int tiles_per_stream;
    int mt32_per_g;
    int span;             // (taps-1)*dilation history rows in front of a tile
    int ksteps;           // 16-k chunks per m-tile in the packed weights (K padded to a multiple of 64)
    unsigned w_bytes, in_bytes;
    unsigned buf_bytes;   // one row buffer
    int n_items;          // batch * tiles_per_stream * groups
    int* err;
};

template <int ACT>
__device__ __forceinline__ float rp_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

__device__ __forceinline__ f16x8p rp_as_f16x8(const u32x4p& v) {
    union { u32x4p u; f16x8p h; } c; c.u = v; return c.h;
}

struct RpItem { int g, b, t0, tcur, n_tiles; };

template <int C, int ACT, int TAPS, int NW>
__global__ __launch_bounds__(64 * NW, 1) void conv_rp16_kernel(ConvArgs a, RpArgs rp) {
    constexpr int RS = 4 * C + 16;                     // LDS row stride in bytes: [C halfs hi][C halfs lo][16 B pad]
    constexpr int CH = C / 16;                         // 16-k chunks per tap
    constexpr int STEPS = TAPS * CH;
    constexpr int NT = 64 * NW;
    constexpr int C8 = C / 8;
    constexpr int PF = 8;                              // weight prefetch distance in 16-k chunks: one wave per SIMD has nobody to hide an L2 round trip behind (PF = 2: 50.8 us on the 64-channel layer, conv_rl16 35.1)
    constexpr int SI = CH;                             // one staging piece per tap of the k loop
    extern __shared__ __attribute__((aligned(16))) unsigned char xs_all[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int G = (int)gridDim.x;

    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, rp.w_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, rp.in_bytes, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int m_tiles = rp.mt32_per_g;

    auto decode = [&](int w) {
        RpItem c;
        c.g = w % a.groups;
        const int rest = w / a.groups;
        const int tile = rest % rp.tiles_per_stream;
        c.b = rest / rp.tiles_per_stream;
        c.t0 = tile * rp.tt;
        c.tcur = min(rp.tt, a.t_out - c.t0);
        c.n_tiles = (c.tcur + 31) >> 5;
        return c;
    };
    // staging piece p of this thread: 8 channels of row (tid + p * NT) / C8 of rows [t0 - span, t0 + 32 * n_tiles); rows past
    // the valid ones read out of bounds (= 0, act(0) = 0: zero rows, as conv_rl16 stages them)
    auto piece_load = [&](const RpItem& c, int p, u32x4p& u, u32x4p& v) {
        const int i = tid + p * NT;
        const int rr = i / C8, c8 = i - rr * C8;
        int row = a.in_row0 + c.t0 + rr;
        row %= a.in_rows;
        const unsigned off = rr < rp.span + c.tcur
            ? (unsigned)((((size_t)c.b * a.in_rows + row) * a.in_ch + a.in_choff + c.g * a.in_gstride + 8 * c8) * 4u) : 0x80000000u;
        u = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, off, 0, 0);
        v = __builtin_amdgcn_raw_buffer_load_b128(rsrc_in, off, 16, 0);
    };
    auto piece_store = [&](unsigned char* xs, const RpItem& c, int p, const u32x4p& u, const u32x4p& v) {
        const int i = tid + p * NT;
        if (i >= (rp.span + 32 * c.n_tiles) * C8) return;
        const int rr = i / C8, c8 = i - rr * C8;
        const float x[8] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w),
                            __uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
        f16x8p hi, lo;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float y = rp_act<ACT>(x[j], a.slope);
            const _Float16 h = (_Float16)y;
            hi[j] = h;
            lo[j] = (_Float16)((y - (float)h) * kRpLoScale);
        }
        unsigned char* d = xs + rr * RS + 16 * c8;
        *reinterpret_cast<f16x8p*>(d) = hi;
        *reinterpret_cast<f16x8p*>(d + 2 * C) = lo;
    };
    auto pieces_of = [&](const RpItem& c) { return ((rp.span + 32 * c.n_tiles) * C8 + NT - 1) / NT; };

    int w = (int)blockIdx.x;
    if (w >= rp.n_items) return;
    RpItem cur = decode(w);
    // ---- prologue: the first item's rows, two pieces in flight ----
    {
        const int np = pieces_of(cur);
        u32x4p u0, v0, u1, v1;
        piece_load(cur, 0, u0, v0);
        for (int p = 0; p < np; p += 2) {
            if (p + 1 < np) piece_load(cur, p + 1, u1, v1);
            piece_store(xs_all, cur, p, u0, v0);
            if (p + 2 < np) piece_load(cur, p + 2, u0, v0);
            if (p + 1 < np) piece_store(xs_all, cur, p + 1, u1, v1);
        }
    }
    __syncthreads();

    bool bad = false;
#if ADK_RP16_DBG & 1
    const bool tr_on = (blockIdx.x == 0 || blockIdx.x == 100) && wave == 0;
    unsigned long long* tr_base = g_rp_trace + (blockIdx.x == 0 ? 0 : 16 * 8);
#endif
    for (int it = 0;; ++it) {
        RP_STAMP(0);
        const int wn = w + G;
        const bool has_next = wn < rp.n_items;
        const RpItem nxt = has_next ? decode(wn) : cur;
        unsigned char* xs = xs_all + (size_t)(it & 1) * rp.buf_bytes;
        unsigned char* xn = xs_all + (size_t)((it & 1) ^ 1) * rp.buf_bytes;
        const int n_pieces = has_next ? pieces_of(nxt) : 0;
        int sp = 0;                                    // staging pieces of the next item done so far
        u32x4p su, sv;
        su = sv = u32x4p{0u, 0u, 0u, 0u};
        if (n_pieces > 0) piece_load(nxt, 0, su, sv);
        auto stage_step = [&]() {
            if (sp < n_pieces) {
                piece_store(xn, nxt, sp, su, sv);
                ++sp;
                if (sp < n_pieces) piece_load(nxt, sp, su, sv);
            }
        };

        const int n_pairs = (cur.n_tiles + 1) >> 1;
        const int items = m_tiles * n_pairs;
        const int rot = (wave + w) % NW;
        for (int item = rot; item < items; item += NW) {
            const int mt = item / n_pairs, nt0 = 2 * (item - mt * n_pairs);
            const bool two = nt0 + 1 < cur.n_tiles;
            const unsigned wbase = (unsigned)((cur.g * m_tiles + mt) * rp.ksteps) * 2048u;
            f32x16 m0, m1, c0, c1;                      // main / cross-term accumulators of the two n-tiles
#pragma unroll
            for (int e = 0; e < 16; ++e) { m0[e] = 0.f; m1[e] = 0.f; c0[e] = 0.f; c1[e] = 0.f; }
            const unsigned char* x0 = xs + (nt0 * 32 + l31) * RS + 16 * lh;
            const unsigned char* x1 = x0 + 32 * RS;
            u32x4p ah[PF + 1], al[PF + 1];
#pragma unroll
            for (int s = 0; s < PF && s < STEPS; ++s) {
                ah[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, wbase + (unsigned)s * 2048u, 0);
                al[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16 + 1024u, wbase + (unsigned)s * 2048u, 0);
            }
            // One wave per SIMD: nobody else hides an LDS round trip, so the B fragments of step s + 1 are read before the MFMAs
            // of step s, and the one- / two-n-tile cases are separate straight-line loops (a branch per step otherwise)
            auto kloop = [&](auto two_c) {
                constexpr bool TWO = decltype(two_c)::value;
                f16x8p bq[2][4];                         // [parity][b0h, b0l, b1h, b1l]
                auto read_b = [&](int s, f16x8p (&b)[4]) {
                    const int tap = s / CH, ch = s - tap * CH;
                    const int off = tap * a.dilation * RS + 32 * ch;
                    b[0] = *reinterpret_cast<const f16x8p*>(x0 + off);
                    b[1] = *reinterpret_cast<const f16x8p*>(x0 + off + 2 * C);
                    if (TWO) {
                        b[2] = *reinterpret_cast<const f16x8p*>(x1 + off);
                        b[3] = *reinterpret_cast<const f16x8p*>(x1 + off + 2 * C);
                    }
                };
                read_b(0, bq[0]);
#pragma unroll
                for (int s = 0; s < STEPS; ++s) {
                    if (s + PF < STEPS) {
                        const unsigned wo = (ADK_RP16_DBG & 2) ? 0u : (unsigned)(s + PF) * 2048u;      // knock-out: one L1-resident fragment
                        ah[(s + PF) % (PF + 1)] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, wbase + wo, 0);
                        al[(s + PF) % (PF + 1)] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16 + 1024u, wbase + wo, 0);
                    }
                    if (s + 1 < STEPS) read_b(s + 1, bq[(s + 1) & 1]);
                    const f16x8p Ah = rp_as_f16x8(ah[s % (PF + 1)]), Al = rp_as_f16x8(al[s % (PF + 1)]);
                    const f16x8p (&b)[4] = bq[s & 1];
                    if (TWO) {
                        m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b[0], m0, 0, 0, 0);
                        m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b[2], m1, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b[1], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b[3], c1, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b[0], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b[2], c1, 0, 0, 0);
                    } else {
                        m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b[0], m0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b[1], c0, 0, 0, 0);
                        c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b[0], c0, 0, 0, 0);
                    }
                    if (s % SI == SI - 1) stage_step();  // the next item's rows, one piece per tap
                }
            };
            if (two) kloop(std::true_type()); else kloop(std::false_type());
            RP_STAMP(1);
            // ---- epilogue (as conv_rl16): acc0 + acc1/2048, bias, residual, output activation, store; non-finite check ----
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1 && !two) break;
                const f32x16& am = j ? m1 : m0;
                const f32x16& ac = j ? c1 : c0;
                const int t = cur.t0 + (nt0 + j) * 32 + l31;
                if (t >= cur.t0 + cur.tcur) continue;
                const float* resp = nullptr;
                if (a.res) {
                    int rrow = a.res_cursor + t;
                    if (rrow >= a.res_rows) rrow -= a.res_rows;
                    resp = a.res + ((size_t)cur.b * a.res_rows + rrow) * a.res_ch + a.res_choff + cur.g * a.res_gstride;
                }
                int orow = a.out_cursor + t;
                if (orow >= a.out_rows) orow -= a.out_rows;
                float* outp = a.out + ((size_t)cur.b * a.out_rows + orow) * a.out_ch + a.out_choff + cur.g * a.cout_g;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int ml = mt * 32 + 8 * qd + 4 * lh;
                    if (ml >= a.cout_g) continue;
                    float4 v = make_float4(fmaf(ac[4 * qd], kRpLoInv, am[4 * qd]), fmaf(ac[4 * qd + 1], kRpLoInv, am[4 * qd + 1]),
                                           fmaf(ac[4 * qd + 2], kRpLoInv, am[4 * qd + 2]), fmaf(ac[4 * qd + 3], kRpLoInv, am[4 * qd + 3]));
                    bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
                    if (a.bias) {
                        const float4 bb = *reinterpret_cast<const float4*>(a.bias + cur.g * a.cout_g + ml);
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
        }
        RP_STAMP(2);
        while (sp < n_pieces) stage_step();             // waves without an item, or more pieces than taps
        RP_STAMP(3);
        if (!has_next) break;
        __syncthreads();                                // the next item's rows are complete; every wave is done with this buffer
        RP_STAMP(4);
        w = wn;
        cur = nxt;
    }
    if (bad) atomicOr(rp.err, 8);
}

int g_rp_enable = -1;       // ADK_CONV_RP16: 1 = take this kernel where conv_rp16_pick says so, 0 = never

template <int C, int NW>
int launch_rp(const ConvArgs& a, hipStream_t s, int tt) {
    RpArgs rp;
    rp.span = (a.taps - 1) * a.dilation;
    rp.mt32_per_g = a.cout_g / 32;
    rp.ksteps = (a.ktot + 63) / 64 * 4;
    rp.w_bytes = (unsigned)((unsigned long long)a.groups * rp.mt32_per_g * rp.ksteps * 2048ull);
    rp.in_bytes = (unsigned)((unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull);
    rp.err = flags_word();
    rp.tt = tt;
    rp.tiles_per_stream = (a.t_out + tt - 1) / tt;
    constexpr int RS = 4 * C + 16;
    const int tt_pad = (std::min(tt, a.t_out) + 31) / 32 * 32;
    rp.buf_bytes = (unsigned)((rp.span + tt_pad) * RS);
    const size_t lds = 2ull * rp.buf_bytes;
    const long long n_items = (long long)a.batch * rp.tiles_per_stream * a.groups;
    if (n_items > 0x7fffffffLL || lds > 160 * 1024) return fail(ADK_ERR_SHAPE, "conv: pipelined rows kernel does not fit");
    rp.n_items = (int)n_items;
    const long long per_cu = std::max<long long>(1, std::min<long long>(160 * 1024 / (long long)lds, 2048 / (64 * NW)));   // resident workgroups per CU
    const unsigned grid = (unsigned)std::min<long long>(n_items, 256 * per_cu);
    auto go = [&](auto kern) -> int {
        static bool attr_set_dev[kMaxDevices] = {};      // per instantiation and device
        bool& attr_set = attr_set_dev[current_device()];
        if (!attr_set) {
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(64 * NW), lds, s, a, rp);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    if (a.act_in == ADK_ACT_LEAKY) return go(conv_rp16_kernel<C, ADK_ACT_LEAKY, 11, NW>);
    if (a.act_in == ADK_ACT_NONE) return go(conv_rp16_kernel<C, ADK_ACT_NONE, 11, NW>);
    return fail(ADK_ERR_ARG, "conv: unsupported input activation for the pipelined rows kernel");
}

int g_rp_buf_kb = -1;       // ADK_RP16_BUF_KB: LDS per row buffer (default 80: one workgroup per CU; 40: two, if the tiles allow)

// time tile: the whole call of a stream when a buffer holds it, else the call cut into equal tiles (multiples of 32) that fit
int rp16_time_tile(const ConvArgs& a) {
    if (g_rp_buf_kb < 0) { const char* e = getenv("ADK_RP16_BUF_KB"); g_rp_buf_kb = (e && atoi(e) >= 8 && atoi(e) <= 80) ? atoi(e) : 80; }
    const int rs = 4 * a.cin_g + 16;
    const int span = (a.taps - 1) * a.dilation;
    const int tt_max = ((g_rp_buf_kb * 1024 / rs - span) / 32) * 32;
    if (tt_max < 32) return 0;
    if (tt_max >= a.t_out) return a.t_out;
    const int tiles = (a.t_out + tt_max - 1) / tt_max;
    return std::min(tt_max, ((a.t_out + tiles - 1) / tiles + 31) / 32 * 32);
}
}  // namespace

// K11 layers the rows kernel supports, with enough items that every CU gets at least two (else conv_rl16's one round is as good)
bool conv_rp16_pick(const ConvArgs& a, bool force) {
    if (g_rp_enable < 0) { const char* e = getenv("ADK_CONV_RP16"); g_rp_enable = e ? atoi(e) : 0; }
    if (!g_rp_enable && !force) return false;
    if (!conv_rl16_supported(a) || a.taps != 11 || a.up != 1) return false;
    if (a.act_in != ADK_ACT_LEAKY && a.act_in != ADK_ACT_NONE) return false;       // the K11 layers of the path are the vocoder's (LeakyReLU)
    if ((unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull >= 0x80000000ull) return false;       // 32-bit byte offsets of the staging loads
    const int tt = rp16_time_tile(a);
    if (tt <= 0) return false;
    const long long n_items = (long long)a.batch * ((a.t_out + tt - 1) / tt) * a.groups;
    return force || n_items >= 512;
}

int launch_conv_rp16(const ConvArgs& a, hipStream_t s) {
    if (a.n_total == 0) return ADK_OK;
    const int tt = rp16_time_tile(a);
    if (tt <= 0) return fail(ADK_ERR_SHAPE, "conv: history too long for the pipelined rows kernel");
    // waves = work items of a workgroup (m-tiles x pairs of n-tiles), 4 or 5
    const int n_tiles = (std::min(tt, a.t_out) + 31) / 32;
    const int items = (a.cout_g / 32) * ((n_tiles + 1) / 2);
    const bool five = items % 5 == 0;
    if (a.cin_g == 32) return five ? launch_rp<32, 5>(a, s, tt) : launch_rp<32, 4>(a, s, tt);
    return five ? launch_rp<64, 5>(a, s, tt) : launch_rp<64, 4>(a, s, tt);
}

}  // namespace adk
