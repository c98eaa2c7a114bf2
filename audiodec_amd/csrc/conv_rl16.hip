// "Rows in LDS" causal conv with SPLIT-f16 operands on the f16 matrix cores (opt-in, ADK_IMPL_SPLIT16 / ADK_IMPL_SPLIT16_ROWS).
//
// Same structure as conv_rl.hip (one workgroup = one (stream, group, time tile), rows + history staged once),
// but every f32 operand is carried as two f16 numbers
//     v = hi + lo / 2048,   hi = f16(v),   lo = f16((v - hi) * 2048)          (22+ significant bits)
// and a product sum is formed from three v_mfma_f32_32x32x16_f16 per 16 k:
//     acc0 += A_hi * B_hi          acc1 += A_hi * B_lo + A_lo * B_hi          result = acc0 + acc1 / 2048
// f16 x f16 products are exact in the f32 accumulator, so the only errors are the dropped lo*lo term
// (2^-22 relative) and the 2^-22 representation error of each operand -- measured on gfx950 against fp64
// (tools/mfma_f16_probe.hip, profiles/r1_f16_split_probe.txt): max |err| / sum|a b| = 8e-8 for K = 352..2816,
// BELOW the 2e-7 of the f32 MFMA chain (which rounds after every product), at 3/16 of its matrix-core time.
// f16 subnormal inputs are honoured by the instruction (same probe); |v| > 65504 overflows hi to inf, which makes every
// output it feeds non-finite -- the epilogue tests its outputs and raises device flag bit 3 (adk_debug_flags).  Weights come pre-split in fragment order (adk_pack_weights_split16).
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

#ifndef ADK_RL16_DBG
#define ADK_RL16_DBG 0      // tuning experiments only: 1 = reuse the first weight fragment, 2 = no epilogue stores, 4 = no MFMA, 8 = no staging loads, 16 = per-workgroup wall clocks
#endif

namespace adk {

#if ADK_RL16_DBG & 16
// per-workgroup wall-clock stamps (s_memrealtime, 100 MHz): entry, rows staged, MFMAs done, exit (tools/kbench prints them)
__device__ unsigned long long g_rl_wg_trace[2048 * 4];
extern "C" int adk_debug_rl_wg_trace(unsigned long long* out, int n) {
    if (n > 2048 * 4) n = 2048 * 4;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rl_wg_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));

struct Rl16Args {
    int tt;               // time-tile length (output steps per workgroup)
    int tiles_per_stream;
    int mt32_per_g;
    int span;             // (taps-1)*dilation history rows in front of a tile
    int ksteps;           // 16-k chunks per m-tile in the packed weights (K padded to a multiple of 64)
    unsigned w_bytes;
    int* err;             // sticky device flags
    // FUSE: the 1x1 conv of a residual unit that consumes this conv's output (residual_unit.py:78-81: x + conv2(act(conv1(act(x)))))
    const float* w2;      // its split16 fragments ([1 group][C/32 m-tiles][C/16 chunks][hi|lo][64][8 halfs])
    const float* bias2;   // or nullptr
    float* out2; int out2_rows, out2_ch, out2_cursor, out2_choff;
    const float* res2; int res2_rows, res2_ch, res2_cursor, res2_choff;
    int act_out2;
};

constexpr float kLoScale = 2048.f, kLoInv = 1.f / 2048.f;

template <int ACT>
__device__ __forceinline__ float rl16_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

__device__ __forceinline__ void split8(const float4& u, const float4& v, f16x8& hi, f16x8& lo) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const _Float16 h = (_Float16)x[j];
        hi[j] = h;
        lo[j] = (_Float16)((x[j] - (float)h) * kLoScale);
    }
}

__device__ __forceinline__ f16x8 as_f16x8(const u32x4s& v) {
    union { u32x4s u; f16x8 h; } c; c.u = v; return c.h;
}

// NW waves per workgroup; each wave owns work items (m-tile, pair of consecutive n-tiles) and keeps each
// weight fragment in registers for both n-tiles.  PF = weight prefetch distance in 16-k chunks.
// FUSE = true: residual unit in one launch.  Phase 1 is the K-tap conv exactly as below, but its result (h) stays in registers;
// after a barrier (every wave is done reading the staged rows) each lane writes act(h), split into f16 hi / lo, over the staged
// rows -- the layout the 1x1 conv's B fragments are read from -- and after a second barrier the same wave runs the 1x1 conv for
// the same (m-tile, n-tile pair), adds bias / residual x and stores.  h never goes to memory (the unfused path writes and
// re-reads 4 bytes per element per stream) and one launch disappears.  Needs at most one work item per wave (the launch
// checks); operations and their order per output element are those of the two separate kernels, so the result is bit-identical.
template <int C, int ACT, int TAPS, int NW, int PF, bool FUSE = false>
__global__ __launch_bounds__(64 * NW, ((PF > 4 || PF >= TAPS * C / 16) ? 2 : 3)) void conv_rl16_kernel(ConvArgs a, Rl16Args rl) {   // all-weights-up-front variants: 256 VGPRs
    constexpr int RS = 4 * C + 16;                     // LDS row stride in bytes: [C halfs hi][C halfs lo][16 B pad]
    constexpr int CH = C / 16;                         // 16-k chunks per tap
    constexpr int STEPS = TAPS * CH;
    constexpr int NT = 64 * NW;
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
#if ADK_RL16_DBG & 16
    const unsigned long long tr0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long tr1 = 0, tr2 = 0;
#endif

    const int g = blockIdx.x % a.groups;
    const int rest = blockIdx.x / a.groups;
    const int tile = rest % rl.tiles_per_stream;
    const int b = rest / rl.tiles_per_stream;
    const int t0 = tile * rl.tt;
    const int tcur = min(rl.tt, a.t_out - t0);
    const int n_tiles = (tcur + 31) >> 5;

    // ---- work items of this wave, and the weight fragments of its FIRST item: issued before the rows are staged, so the
    // L2 round trip of the weights overlaps the one of the rows (for short K -- PF >= STEPS -- that is all the weights) ----
    const int m_tiles = rl.mt32_per_g;
    const int n_pairs = (n_tiles + 1) >> 1;
    const int items = m_tiles * n_pairs;
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, rl.w_bytes, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    const int rot = (wave + blockIdx.x + (blockIdx.x >> 8)) % NW;
    u32x4s ah[PF + 1], al[PF + 1];
    auto preload = [&](unsigned wb) {
#pragma unroll
        for (int s = 0; s < PF && s < STEPS; ++s) {
            ah[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, wb + (unsigned)s * 2048u, 0);
            al[s] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16 + 1024u, wb + (unsigned)s * 2048u, 0);
        }
    };
    const int first_mt = rot / n_pairs;
    if (rot < items) preload((unsigned)((g * m_tiles + first_mt) * rl.ksteps) * 2048u);
    // bias of this wave's first item: fetched now, not in the epilogue (one less dependent round trip before the stores)
    // (short-K variants only: the long-K ones have no registers to spare)
    constexpr bool EARLY_BIAS = (PF > 4 || PF >= STEPS);
    float4 bias0[EARLY_BIAS ? 4 : 1];
    if constexpr (EARLY_BIAS) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            bias0[qd] = make_float4(0.f, 0.f, 0.f, 0.f);
            const int ml = first_mt * 32 + 8 * qd + 4 * lh;
            if (a.bias && rot < items && ml < a.cout_g) bias0[qd] = *reinterpret_cast<const float4*>(a.bias + g * a.cout_g + ml);
        }
    }

    // FUSE: all fragments of the 1x1 conv for this wave's m-tile (C/16 chunks), fetched in the same round trip
    constexpr int ST2 = FUSE ? C / 16 : 1;
    u32x4s a2h[ST2], a2l[ST2];
    if constexpr (FUSE) {
        constexpr int KS2 = (C + 63) / 64 * 4;         // chunks per m-tile in the packed layout (K padded to a multiple of 64)
        const __amdgpu_buffer_rsrc_t rsrc_w2 = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(rl.w2), 0, (unsigned)((C / 32) * KS2 * 2048), 0x00020000);
        const unsigned wb2 = (unsigned)(first_mt * KS2) * 2048u;   // a wave without an item reads past the end: zeros
#pragma unroll
        for (int s2 = 0; s2 < ST2; ++s2) {
            a2h[s2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w2, lane16, wb2 + (unsigned)s2 * 2048u, 0);
            a2l[s2] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w2, lane16 + 1024u, wb2 + (unsigned)s2 * 2048u, 0);
        }
    }

    // ---- stage rows [t0 - span, t0 + 32*n_tiles): activation, split into hi / lo halves ----
    {
        const float* xin = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff + g * a.in_gstride;
        const int rows_valid = rl.span + tcur;
        const int rows_all = rl.span + 32 * n_tiles;
        constexpr int C8 = C / 8;
        const int items = rows_all * C8;
        constexpr int SB = (PF > 4 || PF >= STEPS) ? 4 : 2;     // items per pass: their loads are in flight together (the short-K
                                                                // variants have the registers to take a whole tile in one round trip)
        for (int i0 = tid; i0 < items; i0 += SB * NT) {
            float4 u[SB], v[SB];
            int rr[SB], c8[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int i = i0 + k * NT;
                rr[k] = i / C8; c8[k] = i - rr[k] * C8;
                u[k] = v[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if (i < items && rr[k] < rows_valid && (!(ADK_RL16_DBG & 8) || a.slope == 1.2345e-30f)) {
                    int row = a.in_row0 + t0 + rr[k];
                    row %= a.in_rows;
                    const float4* p = reinterpret_cast<const float4*>(xin + (size_t)row * a.in_ch + 8 * c8[k]);
                    u[k] = p[0]; v[k] = p[1];
                }
            }
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                if (i0 + k * NT >= items) break;
                float4 uu = u[k], vv = v[k];
                uu.x = rl16_act<ACT>(uu.x, a.slope); uu.y = rl16_act<ACT>(uu.y, a.slope);
                uu.z = rl16_act<ACT>(uu.z, a.slope); uu.w = rl16_act<ACT>(uu.w, a.slope);
                vv.x = rl16_act<ACT>(vv.x, a.slope); vv.y = rl16_act<ACT>(vv.y, a.slope);
                vv.z = rl16_act<ACT>(vv.z, a.slope); vv.w = rl16_act<ACT>(vv.w, a.slope);
                f16x8 hi, lo;
                split8(uu, vv, hi, lo);
                unsigned char* d = xs + rr[k] * RS + 16 * c8[k];
                *reinterpret_cast<f16x8*>(d) = hi;
                *reinterpret_cast<f16x8*>(d + 2 * C) = lo;
            }
        }
    }
    __syncthreads();
#if ADK_RL16_DBG & 16
    tr1 = __builtin_amdgcn_s_memrealtime();
#endif

    f32x16 m0, m1, c0, c1;                              // main / cross-term accumulators of the two n-tiles
    int mt = 0, nt0 = 0;
    bool two = false;
    for (int item = rot; item < items; item += NW) {
        mt = item / n_pairs; nt0 = 2 * (item - mt * n_pairs);
        two = nt0 + 1 < n_tiles;
        const unsigned wbase = (unsigned)((g * m_tiles + mt) * rl.ksteps) * 2048u;
#pragma unroll
        for (int e = 0; e < 16; ++e) { m0[e] = 0.f; m1[e] = 0.f; c0[e] = 0.f; c1[e] = 0.f; }
        const unsigned char* x0 = xs + (nt0 * 32 + l31) * RS + 16 * lh;
        const unsigned char* x1 = x0 + 32 * RS;

        if (item != rot) preload(wbase);
#pragma unroll
        for (int s = 0; s < STEPS; ++s) {
            if (s + PF < STEPS && !(ADK_RL16_DBG & 1)) {
                ah[(s + PF) % (PF + 1)] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, wbase + (unsigned)(s + PF) * 2048u, 0);
                al[(s + PF) % (PF + 1)] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16 + 1024u, wbase + (unsigned)(s + PF) * 2048u, 0);
            }
            const int tap = s / CH, ch = s - tap * CH;
            const int off = tap * a.dilation * RS + 32 * ch;
            const f16x8 Ah = as_f16x8(ah[s % (PF + 1)]), Al = as_f16x8(al[s % (PF + 1)]);
            const f16x8 b0h = *reinterpret_cast<const f16x8*>(x0 + off);
            const f16x8 b0l = *reinterpret_cast<const f16x8*>(x0 + off + 2 * C);
            if (ADK_RL16_DBG & 4) {
                m0[0] += (float)b0h[0] + (float)b0l[1] + (float)Ah[0] + (float)Al[0];
            } else if (two) {
                const f16x8 b1h = *reinterpret_cast<const f16x8*>(x1 + off);
                const f16x8 b1l = *reinterpret_cast<const f16x8*>(x1 + off + 2 * C);
                m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0h, m0, 0, 0, 0);
                m1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b1h, m1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0l, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b1l, c1, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b0h, c0, 0, 0, 0);
                c1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b1h, c1, 0, 0, 0);
            } else {
                m0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0h, m0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0l, c0, 0, 0, 0);
                c0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b0h, c0, 0, 0, 0);
            }
        }
#if ADK_RL16_DBG & 16
        __builtin_amdgcn_sched_barrier(0);
        tr2 = __builtin_amdgcn_s_memrealtime();
        __builtin_amdgcn_sched_barrier(0);
#endif
        if constexpr (FUSE) break;                      // at most one item per wave; phase 2 below consumes the accumulators
        // ---- epilogue: acc0 + acc1/2048, bias, residual, output activation, store.  An operand beyond the f16 range was
        // split into inf parts, so every output it feeds is non-finite: checked here, once per output ----
        bool bad = false;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (j == 1 && !two) break;
            const f32x16& am = j ? m1 : m0;
            const f32x16& ac = j ? c1 : c0;
            const int t = t0 + (nt0 + j) * 32 + l31;
            if (t >= t0 + tcur) continue;
            const float* resp = nullptr;
            if (a.res) {
                int rrow = a.res_cursor + t;
                if (rrow >= a.res_rows) rrow -= a.res_rows;
                resp = a.res + ((size_t)b * a.res_rows + rrow) * a.res_ch + a.res_choff + g * a.res_gstride;
            }
            int orow = a.out_cursor + t * a.up;              // up > 1: polyphase transposed conv, GEMM row m = phase * cout_real + co
            if (orow >= a.out_rows) orow -= a.out_rows;
            float* outb = a.out + (size_t)b * a.out_rows * a.out_ch + a.out_choff + g * a.cout_g;
            float* outp = outb + (size_t)orow * a.out_ch;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = mt * 32 + 8 * qd + 4 * lh;
                if (ml >= a.cout_g) continue;
                float4 v = make_float4(fmaf(ac[4 * qd], kLoInv, am[4 * qd]), fmaf(ac[4 * qd + 1], kLoInv, am[4 * qd + 1]),
                                       fmaf(ac[4 * qd + 2], kLoInv, am[4 * qd + 2]), fmaf(ac[4 * qd + 3], kLoInv, am[4 * qd + 3]));
                bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
                if (a.bias) {
                    const float4 bb = (EARLY_BIAS && item == rot) ? bias0[EARLY_BIAS ? qd : 0] : *reinterpret_cast<const float4*>(a.bias + g * a.cout_g + ml);
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
                float* dst = outp + ml;
                if (a.up > 1) {
                    const int ph = ml / a.cout_real;
                    int r2 = orow + ph;
                    if (r2 >= a.out_rows) r2 -= a.out_rows;
                    dst = outb + (size_t)r2 * a.out_ch + (ml - ph * a.cout_real);
                }
                if (!(ADK_RL16_DBG & 2) || v.x == 1.2345e-30f) *reinterpret_cast<float4*>(dst) = v;
            }
        }
        if (bad) atomicOr(rl.err, 8);
    }
    if constexpr (FUSE) {
        const bool has_item = rot < items;
        bool bad = false;
        __syncthreads();                                // every wave is done reading the staged rows of phase 1
        if (has_item) {
            // h = acc0 + acc1/2048 (+ bias): what the unfused kernel stores; act(h), split, goes over the staged rows [0, 32*n_tiles)
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1 && !two) break;
                const f32x16& am = j ? m1 : m0;
                const f32x16& ac = j ? c1 : c0;
                unsigned char* row = xs + ((nt0 + j) * 32 + l31) * RS;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int ml = mt * 32 + 8 * qd + 4 * lh;
                    float h[4] = {fmaf(ac[4 * qd], kLoInv, am[4 * qd]), fmaf(ac[4 * qd + 1], kLoInv, am[4 * qd + 1]),
                                  fmaf(ac[4 * qd + 2], kLoInv, am[4 * qd + 2]), fmaf(ac[4 * qd + 3], kLoInv, am[4 * qd + 3])};
                    bad |= !(fabsf(h[0]) <= 3.0e38f) | !(fabsf(h[1]) <= 3.0e38f) | !(fabsf(h[2]) <= 3.0e38f) | !(fabsf(h[3]) <= 3.0e38f);
                    if (a.bias) {
                        const float4 bb = *reinterpret_cast<const float4*>(a.bias + g * a.cout_g + ml);
                        h[0] += bb.x; h[1] += bb.y; h[2] += bb.z; h[3] += bb.w;
                    }
                    typedef _Float16 f16x4r __attribute__((ext_vector_type(4)));
                    f16x4r hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = rl16_act<ACT>(h[e], a.slope);
                        const _Float16 hh = (_Float16)y;
                        hi[e] = hh;
                        lo[e] = (_Float16)((y - (float)hh) * kLoScale);
                    }
                    *reinterpret_cast<f16x4r*>(row + 2 * ml) = hi;
                    *reinterpret_cast<f16x4r*>(row + 2 * C + 2 * ml) = lo;
                }
            }
        }
        __syncthreads();                                // act(h) of all channels of this wave's time steps is in LDS
        if (has_item) {
            f32x16 q0, q1, r0, r1;
#pragma unroll
            for (int e = 0; e < 16; ++e) { q0[e] = 0.f; q1[e] = 0.f; r0[e] = 0.f; r1[e] = 0.f; }
            const unsigned char* x0 = xs + (nt0 * 32 + l31) * RS + 16 * lh;
            const unsigned char* x1 = x0 + 32 * RS;
#pragma unroll
            for (int s2 = 0; s2 < ST2; ++s2) {          // the 1x1 conv: same MFMA sequence as conv_rl16_kernel<C, ACT, 1, ...>
                const f16x8 Ah = as_f16x8(a2h[s2]), Al = as_f16x8(a2l[s2]);
                const f16x8 b0h = *reinterpret_cast<const f16x8*>(x0 + 32 * s2);
                const f16x8 b0l = *reinterpret_cast<const f16x8*>(x0 + 32 * s2 + 2 * C);
                if (two) {
                    const f16x8 b1h = *reinterpret_cast<const f16x8*>(x1 + 32 * s2);
                    const f16x8 b1l = *reinterpret_cast<const f16x8*>(x1 + 32 * s2 + 2 * C);
                    q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0h, q0, 0, 0, 0);
                    q1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b1h, q1, 0, 0, 0);
                    r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0l, r0, 0, 0, 0);
                    r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b1l, r1, 0, 0, 0);
                    r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b0h, r0, 0, 0, 0);
                    r1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b1h, r1, 0, 0, 0);
                } else {
                    q0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0h, q0, 0, 0, 0);
                    r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, b0l, r0, 0, 0, 0);
                    r0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, b0h, r0, 0, 0, 0);
                }
            }
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                if (j == 1 && !two) break;
                const f32x16& am = j ? q1 : q0;
                const f32x16& ac = j ? r1 : r0;
                const int t = t0 + (nt0 + j) * 32 + l31;
                if (t >= t0 + tcur) continue;
                const float* resp = nullptr;
                if (rl.res2) {
                    int rrow = rl.res2_cursor + t;
                    if (rrow >= rl.res2_rows) rrow -= rl.res2_rows;
                    resp = rl.res2 + ((size_t)b * rl.res2_rows + rrow) * rl.res2_ch + rl.res2_choff;
                }
                int orow = rl.out2_cursor + t;
                if (orow >= rl.out2_rows) orow -= rl.out2_rows;
                float* outp = rl.out2 + ((size_t)b * rl.out2_rows + orow) * rl.out2_ch + rl.out2_choff;
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) {
                    const int ml = mt * 32 + 8 * qd + 4 * lh;
                    float4 v = make_float4(fmaf(ac[4 * qd], kLoInv, am[4 * qd]), fmaf(ac[4 * qd + 1], kLoInv, am[4 * qd + 1]),
                                           fmaf(ac[4 * qd + 2], kLoInv, am[4 * qd + 2]), fmaf(ac[4 * qd + 3], kLoInv, am[4 * qd + 3]));
                    bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
                    if (rl.bias2) {
                        const float4 bb = *reinterpret_cast<const float4*>(rl.bias2 + ml);
                        v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                    }
                    if (resp) {
                        const float4 rr = *reinterpret_cast<const float4*>(resp + ml);
                        v.x += rr.x; v.y += rr.y; v.z += rr.z; v.w += rr.w;
                    }
                    if (rl.act_out2 != ADK_ACT_NONE) {
                        v.x = act_apply(v.x, rl.act_out2, 0.f); v.y = act_apply(v.y, rl.act_out2, 0.f);
                        v.z = act_apply(v.z, rl.act_out2, 0.f); v.w = act_apply(v.w, rl.act_out2, 0.f);
                    }
                    *reinterpret_cast<float4*>(outp + ml) = v;
                }
            }
        }
        if (bad) atomicOr(rl.err, 8);
    }
#if ADK_RL16_DBG & 16
    if (wave == 0 && lane == 0 && blockIdx.x < 2048) {
        unsigned long long* t = g_rl_wg_trace + (size_t)blockIdx.x * 4;
        t[0] = tr0; t[1] = tr1; t[2] = tr2; t[3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
}

// w [groups*cout_g][ktot] row-major (k = tap*cin_g + ci) -> [g][m-tile 32][16-k chunk][hi | lo][lane 64][8 halfs],
// lane (i = lane & 31, h = lane >> 5) holding W[32*mt + i][16*chunk + 8*h + 0..7]; rows beyond cout_g and the K tail
// (K padded to a multiple of 64, like adk_pack_weights_mfma) are zero.
__global__ __launch_bounds__(256) void pack_split16_kernel(const float* __restrict__ w, _Float16* __restrict__ out, int groups,
                                                           int cout_g, int ktot, int* err) {
    const int mt32 = (cout_g + 31) / 32, chunks = (ktot + 63) / 64 * 4;
    const long long total = (long long)groups * mt32 * chunks * 512;     // (hi or lo) half-pairs: one thread per (.., lane, j)
    bool bad = false;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int j = (int)(gid & 7), lane = (int)((gid >> 3) & 63);
        const long long blk = gid >> 9;                                   // (g, mt, chunk)
        const int ch = (int)(blk % chunks);
        const long long gm = blk / chunks;
        const int mt = (int)(gm % mt32), g = (int)(gm / mt32);
        const int i = lane & 31, h = lane >> 5;
        const int m = mt * 32 + i, k = 16 * ch + 8 * h + j;
        float v = 0.f;
        if (m < cout_g && k < ktot) v = w[((size_t)g * cout_g + m) * ktot + k];
        const _Float16 hi = (_Float16)v;
        const _Float16 lo = (_Float16)((v - (float)hi) * kLoScale);
        bad |= fabsf(v) > 65504.f;
        _Float16* o = out + blk * 1024 + lane * 8 + j;
        o[0] = hi; o[512] = lo;
    }
    if (bad) atomicOr(err, 8);
}

int launch_pack_split16(const float* w, float* out, int groups, int cout_g, int ktot, hipStream_t s) {
    const long long total = (long long)groups * ((cout_g + 31) / 32) * ((ktot + 63) / 64 * 4) * 512;
    long long blocks = (total + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(pack_split16_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, reinterpret_cast<_Float16*>(out), groups, cout_g, ktot, flags_word());
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

bool conv_rl16_supported(const ConvArgs& a) {
    if (!a.wfrag || a.stride != 1) return false;
    if (a.cin_g != 32 && a.cin_g != 64) return false;
    if (a.up != 1 && (a.taps != 2 || a.groups != 1 || a.cout_real % 4 || a.res)) return false;   // transposed convs: 2 taps, polyphase rows
    if (a.taps != 1 && a.taps != 2 && a.taps != 3 && a.taps != 7 && a.taps != 11) return false;   // tap loop unrolled at compile time
    if (a.cout_g % 32 != 0) return false;
    if ((a.in_ch % 4) || (a.in_choff % 4) || (a.in_gstride % 4) || (a.out_ch % 4) || (a.out_choff % 4)) return false;
    if (a.res && ((a.res_ch % 4) || (a.res_choff % 4) || (a.res_gstride % 4))) return false;
    if ((reinterpret_cast<uintptr_t>(a.in) | reinterpret_cast<uintptr_t>(a.out) | reinterpret_cast<uintptr_t>(a.wfrag)) & 15) return false;
    if (a.res && (reinterpret_cast<uintptr_t>(a.res) & 15)) return false;
    if (a.bias && (reinterpret_cast<uintptr_t>(a.bias) & 15)) return false;
    return true;
}

static bool rl16_few_streams() {                 // ADK_RL16_FEW=0 (tuning): the round-1 rule -- rows kernel only when >= 192 long tiles
    static int v = -1;
    if (v < 0) { const char* e = getenv("ADK_RL16_FEW"); v = e ? atoi(e) : 1; }
    return v != 0;
}

// Time steps per workgroup.  0: the history does not fit.
static int rl16_time_tile(const ConvArgs& a) {
    const int rs = 4 * a.cin_g + 16;
    const int span = (a.taps - 1) * a.dilation;
    int tt = ((54000 / rs - span) / 32) * 32;
    if (tt < 32) return 0;
    const int nt32 = (a.t_out + 31) / 32;
    if (tt >= a.t_out) tt = a.t_out;
    // few streams: one n-tile per workgroup while that keeps the launch within one round of resident workgroups (3 per CU).
    // Re-staging the history per tile costs nothing when most CUs would idle otherwise (measured, tools/run_r2q.sh: the
    // 32-channel K7 conv at 1 / 32 / 64 streams 9.9 / 10.5 / 10.7 -> 6.4 / 7.1 / 8.5 us; stream-K takes 15 / 17 / 20 us)
    if (rl16_few_streams() && nt32 > 1 && (long long)a.batch * nt32 * a.groups <= 768) tt = 32;
    // one tile would cover the whole call: cut a long one (>= 8 n-tiles, e.g. the 300-step frame of the 32-channel
    // layers) into two balanced halves -- twice the workgroups, two dispatch rounds whose load / matrix-core / store
    // phases overlap instead of running in lockstep (measured 38.0 -> 34.2 us; shorter tiles lose to the re-staged history)
    else if (tt >= a.t_out && nt32 >= 8) tt = ((nt32 + 1) / 2) * 32;
    // ... and a short one whose (m-tile, n-tile pair) items would need two passes of the 4 waves (the 64 -> 32 transposed
    // conv: 3 m-tiles x 2 pairs) when halving it leaves at most one item per wave
    else if (tt >= a.t_out && nt32 >= 4 && (a.cout_g / 32) * ((nt32 + 1) / 2) > 4 && (a.cout_g / 32) * (((nt32 + 1) / 2 + 1) / 2) <= 4)
        tt = ((nt32 + 1) / 2) * 32;
    static int tt_env = -1;                                                  // tuning: shorter time tiles (more, smaller workgroups)
    if (tt_env < 0) { const char* e = getenv("ADK_RL16_TT"); tt_env = e ? atoi(e) : 0; }
    if (tt_env >= 32 && tt_env < tt) tt = tt_env / 32 * 32;
    return tt;
}

// AUTO (ADK_IMPL_SPLIT16): rows-in-LDS wherever the layer qualifies and a call brings at least most of an n-tile per
// stream (measured faster than the stream-K variant at every stream count from 1 to 256, tools/run_r2q.sh)
bool conv_rl16_preferred(const ConvArgs& a) {
    if (!conv_rl16_supported(a) || a.t_out < 24) return false;
    const int tt = rl16_time_tile(a);
    if (tt <= 0) return false;
    if (!rl16_few_streams()) return (long long)a.batch * ((a.t_out + tt - 1) / tt) * a.groups >= 192;
    return true;
}

namespace {
template <int C, int NW>
int launch_rl16(const ConvArgs& a, hipStream_t s, int tt) {
    Rl16Args rl;
    rl.span = (a.taps - 1) * a.dilation;
    rl.mt32_per_g = a.cout_g / 32;
    rl.ksteps = (a.ktot + 63) / 64 * 4;
    rl.w_bytes = (unsigned)((unsigned long long)a.groups * rl.mt32_per_g * rl.ksteps * 2048ull);
    rl.err = conv_err_word(a);
    rl.tt = tt;
    rl.tiles_per_stream = (a.t_out + tt - 1) / tt;
    constexpr int RS = 4 * C + 16;
    const int tt_pad = (std::min(tt, a.t_out) + 31) / 32 * 32;
    const size_t lds = (size_t)(rl.span + tt_pad) * RS;
    const long long blocks = (long long)a.batch * rl.tiles_per_stream * a.groups;
    if (blocks > 0x7fffffffLL) return fail(ADK_ERR_SHAPE, "conv: too many workgroups");
    if (lds > 64 * 1024) return fail(ADK_ERR_SHAPE, "conv: rows-in-LDS tile exceeds 64 KiB");
    auto go = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), lds, s, a, rl);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    // weight prefetch distance in 16-k chunks: 2 for the long-K layers (3 spills at the 168-VGPR budget of 3 workgroups
    // per CU); short K (<= 8 chunks: 1x1 convs, the 2-tap transposed convs) takes ALL its weight fragments up front, in
    // the same round trip as the rows
    constexpr int PF = 2;
    constexpr int S1 = 1 * C / 16, S2 = 2 * C / 16;
    auto by_taps = [&](auto act) -> int {
        constexpr int ACT = decltype(act)::value;
        if (a.taps == 1) return go(conv_rl16_kernel<C, ACT, 1, NW, S1>);
        if (a.taps == 2) return go(conv_rl16_kernel<C, ACT, 2, NW, S2>);
        if (a.taps == 3) return go(conv_rl16_kernel<C, ACT, 3, NW, PF>);
        if (a.taps == 7) return go(conv_rl16_kernel<C, ACT, 7, NW, PF>);
        return go(conv_rl16_kernel<C, ACT, 11, NW, PF>);
    };
    if (a.act_in == ADK_ACT_ELU) return by_taps(std::integral_constant<int, ADK_ACT_ELU>());
    if (a.act_in == ADK_ACT_LEAKY) return by_taps(std::integral_constant<int, ADK_ACT_LEAKY>());
    if (a.act_in == ADK_ACT_NONE) return by_taps(std::integral_constant<int, ADK_ACT_NONE>());
    return fail(ADK_ERR_ARG, "conv: unsupported input activation for the rows-in-LDS kernel");
}
}  // namespace

int launch_conv_rl16(const ConvArgs& a, hipStream_t s) {
    if (a.n_total == 0) return ADK_OK;
    const int tt = rl16_time_tile(a);
    if (tt <= 0) return fail(ADK_ERR_SHAPE, "conv: history too long for the rows-in-LDS kernel");
    // work items of a workgroup = m-tiles x pairs of n-tiles; 5 waves when that is a multiple of 5 (the 300-step
    // frame of a 32-channel layer: 10 n-tiles), else 4
    const int n_tiles = (std::min(tt, a.t_out) + 31) / 32;
    const int items = (a.cout_g / 32) * ((n_tiles + 1) / 2);
    const bool five = items % 5 == 0;
    if (a.cin_g == 32) return five ? launch_rl16<32, 5>(a, s, tt) : launch_rl16<32, 4>(a, s, tt);
    return five ? launch_rl16<64, 5>(a, s, tt) : launch_rl16<64, 4>(a, s, tt);
}


// ---- residual unit in one launch: K-tap conv (a1) -> 1x1 conv with residual (a2), see conv_rl16_kernel<..., FUSE> ----
bool conv_rl16_fusable(const ConvArgs& a1, const ConvArgs& a2) {
    if (!conv_rl16_preferred(a1) || !a2.wfrag) return false;
    if (a1.taps != 7 || a1.up != 1 || a1.groups != 1 || a1.res || a1.act_out != ADK_ACT_NONE) return false;
    if (a2.taps != 1 || a2.up != 1 || a2.groups != 1 || a2.stride != 1) return false;
    if (a2.cin_g != a1.cout_g || a2.cout_g != a1.cout_g || a1.cin_g != a1.cout_g) return false;          // C -> C -> C
    if (a2.act_in != a1.act_in || a2.slope != a1.slope || a2.batch != a1.batch || a2.t_out != a1.t_out) return false;
    if (a2.in != a1.out || a2.in_rows != a1.out_rows || a2.in_ch != a1.out_ch || a2.in_choff != a1.out_choff) return false;   // h feeds only the 1x1
    if ((a2.out_ch % 4) || (a2.out_choff % 4) || (reinterpret_cast<uintptr_t>(a2.out) & 15) || (reinterpret_cast<uintptr_t>(a2.wfrag) & 15)) return false;
    if (a2.res && ((a2.res_ch % 4) || (a2.res_choff % 4) || (reinterpret_cast<uintptr_t>(a2.res) & 15))) return false;
    if (a2.bias && (reinterpret_cast<uintptr_t>(a2.bias) & 15)) return false;
    return true;
}

namespace {
template <int C, int NW>
int launch_rl16_fused(const ConvArgs& a, const ConvArgs& a2, hipStream_t s, int tt) {
    Rl16Args rl;
    rl.span = (a.taps - 1) * a.dilation;
    rl.mt32_per_g = a.cout_g / 32;
    rl.ksteps = (a.ktot + 63) / 64 * 4;
    rl.w_bytes = (unsigned)((unsigned long long)a.groups * rl.mt32_per_g * rl.ksteps * 2048ull);
    rl.err = conv_err_word(a);
    rl.tt = tt;
    rl.tiles_per_stream = (a.t_out + tt - 1) / tt;
    rl.w2 = a2.wfrag; rl.bias2 = a2.bias;
    rl.out2 = a2.out; rl.out2_rows = a2.out_rows; rl.out2_ch = a2.out_ch; rl.out2_cursor = a2.out_cursor; rl.out2_choff = a2.out_choff;
    rl.res2 = a2.res; rl.res2_rows = a2.res_rows; rl.res2_ch = a2.res_ch; rl.res2_cursor = a2.res_cursor; rl.res2_choff = a2.res_choff;
    rl.act_out2 = a2.act_out;
    constexpr int RS = 4 * C + 16;
    const int tt_pad = (std::min(tt, a.t_out) + 31) / 32 * 32;
    const size_t lds = (size_t)(rl.span + tt_pad) * RS;
    const long long blocks = (long long)a.batch * rl.tiles_per_stream;
    if (blocks > 0x7fffffffLL || lds > 64 * 1024) return fail(ADK_ERR_SHAPE, "conv: fused residual unit does not fit");
    auto go = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3((unsigned)blocks), dim3(64 * NW), lds, s, a, rl);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    if (a.act_in == ADK_ACT_ELU) return go(conv_rl16_kernel<C, ADK_ACT_ELU, 7, NW, 2, true>);
    if (a.act_in == ADK_ACT_LEAKY) return go(conv_rl16_kernel<C, ADK_ACT_LEAKY, 7, NW, 2, true>);
    return go(conv_rl16_kernel<C, ADK_ACT_NONE, 7, NW, 2, true>);
}
}  // namespace

// 0 = launched; ADK_ERR_STATE = the pair cannot be fused for this call (the caller launches the two ops separately)
int launch_conv_rl16_fused(const ConvArgs& a, const ConvArgs& a2, hipStream_t s) {
    if (a.n_total == 0) return ADK_OK;
    if (!conv_rl16_fusable(a, a2)) return ADK_ERR_STATE;
    // the same time tiling as launch_conv_rl16 (so that phase 1 is the same launch), then: at most one item per wave
    const int tt = rl16_time_tile(a);
    if (tt <= 0) return ADK_ERR_STATE;
    const int n_tiles = (std::min(tt, a.t_out) + 31) / 32;
    const int items = (a.cout_g / 32) * ((n_tiles + 1) / 2);
    if (items > 5) return ADK_ERR_STATE;
    const bool five = items == 5;
    if (a.cin_g == 32) return five ? launch_rl16_fused<32, 5>(a, a2, s, tt) : launch_rl16_fused<32, 4>(a, a2, s, tt);
    return five ? launch_rl16_fused<64, 5>(a, a2, s, tt) : launch_rl16_fused<64, 4>(a, a2, s, tt);
}

}  // namespace adk
