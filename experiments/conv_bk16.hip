// Big-tile split-f16 implicit-GEMM conv, register-staged: 128 x 128 workgroup tiles, 64 x 64 wave tiles.
//
// Same arithmetic, operands and persistent stream-K schedule as conv_sk_kernel<SPLIT> (conv_mfma.hip); replaces F.conv1d /
// F.conv_transpose1d of CausalConv1d.inference / CausalConvTranspose1d.inference (layers/conv_layer.py:153-156, 194-197) fused
// with input activation, bias and residual add, for the layers with MANY tiles and long K: the grouped K11 convs of vocoder
// stages 0-1 (HiFiGANResidualBlock.inference, modules/residual_block.py:99-105).
//
// Why another tile shape.  Round-2 counters and knock-outs of conv_sk16<64x64> on those layers (profiles/r2_sk16_analysis.md):
// 14 VALU + 5 SALU instructions are issued per MFMA, the matrix cores are busy 13 % of the time, and removing ALL memory traffic
// (no activation loads, one L1-resident weight chunk) only takes the launch from 50 to 38 us: the kernel is bound by the
// instructions AROUND the MFMAs -- staging, conversion, address updates, fragment reads -- which a 32 x 32 wave tile amortises
// over 12 MFMAs per 64-deep chunk.  Here a wave owns 64 x 64 outputs (2 x 2 MFMA tiles: every A and every B fragment feeds two
// MFMA triples) and a workgroup 128 x 128: per chunk a wave issues 48 MFMAs for twice the staging work, i.e. half the
// instructions, a third of the bytes (340 B instead of 1 KiB) and half the conversions per MFMA.  One workgroup per CU (330
// VGPRs per wave), so the schedule uses at most 256 persistent workgroups; a tile cut by a range boundary costs a 64 KB slab of
// partial sums (written and read as contiguous KiB per wave instruction), which is why the dispatch takes this kernel only
// where tiles are plentiful (>= 48) -- with 10-30 tiles over 256 workgroups the owner's serial reduction dominates (measured
// with the LDS-DMA variant of this tile shape, conv_gk16.hip).
//
// MEASURED (tools/kbench, 256 streams, profiles/r2_sk16_analysis.md): correct (max |d| 4-6e-6 vs the exact-f32 kernel) and NOT
// faster -- grouped K11 256-ch 51.6 vs 51.6 us, 128-ch 58.5 vs 51.9 us, encoder K7 128-ch 37.7 vs 28.7 us.  With one workgroup
// of four waves per CU (330 registers per lane) nothing covers the ~2.4 us a dependent round trip to L2 / Infinity Cache takes
// under this load: an iteration is one exposed round trip again, only with more work behind it.  Kept OPT-IN (ADK_CONV_BK16=1,
// ADK_IMPL_SPLIT16_BK) next to conv_gk16 as the second half of that experiment.
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

namespace adk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8b __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4b __attribute__((ext_vector_type(4)));
typedef unsigned u32x4b __attribute__((ext_vector_type(4)));

namespace {
constexpr float kBkLoScale = 2048.f, kBkLoInv = 1.f / 2048.f;
constexpr int BK_KC = 64;                       // K chunk: two 32-channel half-chunks (each one tap x 32 channels)
constexpr int BK_BM = 128, BK_BN = 128;
constexpr int BK_LDK = BK_KC + 4;               // LDS row stride in floats: [64 halfs hi][64 halfs lo][16 B pad] = 272 bytes

struct BkArgs {
    float* ws;            // partial-tile slabs: [G][16 pieces][256 threads][16 B]
    unsigned* flags;      // [G] publish flags (epoch-tagged)
    unsigned epoch;
    int G;
    int m_tiles, n_tiles, nchunks;
    int cpt;              // 32-channel blocks per tap
    int kgroups;          // 1 KiB fragment blocks per 32-row m-tile in the packed weights = 2 * (16-k chunks), K padded to 64
    int mt32_per_g;
    unsigned in_bytes, w_bytes, ws_bytes;
    int* err;
    float inv_t_out;
    long long total;      // tiles * nchunks
};

template <int ACT>
__device__ __forceinline__ float bk_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

__device__ __forceinline__ long long bk_u0(int r, const BkArgs& bk) { return (long long)r * bk.total / bk.G; }

__device__ __forceinline__ int bk_div(int n, int d, float inv_d) {
    int q = (int)(__int2float_rn(n) * inv_d);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; }
    return q;
}

__device__ __forceinline__ float4 bk_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4b v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// bias, residual, output activation, store for one 32 x 64 accumulator block (two n-tiles) of a wave
__device__ __forceinline__ void bk_epilogue(const ConvArgs& a, const f32x16 (&acc)[2], int g, int ml0, int n0w, int lane, bool& bad) {
    const int l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int n = n0w + j * 32 + l31;
        if (n >= a.n_total) continue;
        const int b = n / a.t_out, t = n - b * a.t_out;
        const float* resp = nullptr;
        if (a.res) {
            int rrow = a.res_cursor + t;
            if (rrow >= a.res_rows) rrow -= a.res_rows;
            resp = a.res + ((size_t)b * a.res_rows + rrow) * a.res_ch + a.res_choff + g * a.res_gstride;
        }
        float* outb = a.out + (size_t)b * a.out_rows * a.out_ch + a.out_choff;
        const int obase = a.out_cursor + t * a.up;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int ml = ml0 + 8 * qd + 4 * lh;
            if (ml >= a.cout_g) continue;
            const int mg = g * a.cout_g + ml;
            float4 v = make_float4(acc[j][4 * qd], acc[j][4 * qd + 1], acc[j][4 * qd + 2], acc[j][4 * qd + 3]);
            bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
            if (a.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(a.bias + mg);
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
            int orow = obase, ocol = mg;
            if (a.up > 1) { const int ph = mg / a.cout_real; orow += ph; ocol = mg - ph * a.cout_real; }
            if (orow >= a.out_rows) orow -= a.out_rows;
            *reinterpret_cast<float4*>(outb + (size_t)orow * a.out_ch + ocol) = v;
        }
    }
}

// 4 waves as 2 (M) x 2 (N), each a 64 x 64 output block = 2 x 2 MFMA tiles.  One iteration = one 64-deep K chunk of one tile.
template <int ACT>
__global__ __launch_bounds__(256, 1) void conv_bk16_kernel(ConvArgs a, BkArgs bk) {
    constexpr int QPC = 16;                           // 16-byte pieces per staged column (64 floats)
    constexpr int CPR = 256 / QPC;                    // columns staged per round
    constexpr int RB = BK_BN / CPR;                   // staging rounds: 8
    extern __shared__ __attribute__((aligned(16))) float Bs[];   // [2][BN * LDK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;
    const int srow = tid / QPC, quad = tid % QPC;
    const int half = quad >> 3;                       // which 32-channel sub-chunk of the chunk this thread stages

    const int r = (int)(blockIdx.x & 7) * (bk.G >> 3) + (int)(blockIdx.x >> 3);      // XCD-contiguous ranges (speed only)
    const long long u0 = bk_u0(r, bk), u1 = bk_u0(r + 1, bk);
    if (u0 >= u1) return;

    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, bk.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, bk.w_bytes, 0x00020000);
    const unsigned row_bytes = (unsigned)a.in_ch * 4u;
    const unsigned ring_bytes = (unsigned)a.in_rows * row_bytes;
    const unsigned dil_bytes = (unsigned)a.dilation * row_bytes;
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr unsigned OOB = 0x80000000u;

    // ---- staging state: describes the NEXT chunk to be loaded ----
    int s_tile, s_kc;
    int s_g = 0, s_mt = 0, s_nt = 0;
    unsigned s_wbase[2] = {0, 0};                      // byte offsets of this wave's two fragment streams (OOB-ish if the m-tile does not exist)
    int t_tap, t_cblk;
    unsigned colb[RB], rowb[RB];

    auto tile_coords = [&](int tile, int& g, int& mt, int& nt) {
        mt = tile % bk.m_tiles;
        const int rest = tile / bk.m_tiles;
        nt = rest % bk.n_tiles;
        g = rest / bk.n_tiles;
    };
    auto stage_tile = [&](int tile, int kc0) {
        tile_coords(tile, s_g, s_mt, s_nt);
        s_tile = tile; s_kc = kc0;
        const int j = 2 * kc0 + half;
        t_tap = j / bk.cpt; t_cblk = j - t_tap * bk.cpt;
        const int n0 = s_nt * BK_BN;
        const unsigned tap_bytes = (unsigned)t_tap * dil_bytes;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int n = n0 + srow + CPR * rr;
            const int nn = n < a.n_total ? n : 0;
            const int b = bk_div(nn, a.t_out, bk.inv_t_out), t = nn - b * a.t_out;
            int row = a.in_row0 + t * a.stride;
            if (row >= a.in_rows) row -= a.in_rows;
            unsigned rbv = (unsigned)row * row_bytes + tap_bytes;
            if (rbv >= ring_bytes) rbv -= ring_bytes;
            rowb[rr] = rbv;
            colb[rr] = n < a.n_total ? ((unsigned)b * ring_bytes + (unsigned)(a.in_choff + s_g * a.in_gstride + 4 * (quad & 7)) * 4u) : OOB;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int mtile32 = s_mt * (BK_BM / 32) + 2 * wm + i;
            s_wbase[i] = mtile32 < bk.mt32_per_g ? (unsigned)((s_g * bk.mt32_per_g + mtile32) * bk.kgroups) * 1024u : 0xfff00000u;
        }
    };
    auto stage_advance = [&]() {
        ++s_kc;
        if (s_kc == bk.nchunks) { stage_tile(s_tile + 1, 0); return; }
        t_cblk += 2;
        while (t_cblk >= bk.cpt) {
            t_cblk -= bk.cpt; ++t_tap;
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                unsigned rbv = rowb[rr] + dil_bytes;
                if (rbv >= ring_bytes) rbv -= ring_bytes;
                rowb[rr] = rbv;
            }
        }
    };

    float4 rb[RB];
    float4 a_nxt[2][8];
    auto gload = [&]() {
        const bool k_ok = t_tap < a.taps;              // false on the zero-padded K tail
        const unsigned cb = (unsigned)t_cblk * 128u;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const unsigned vo = (k_ok && colb[rr] != OOB) ? colb[rr] + rowb[rr] + cb : OOB;
            rb[rr] = bk_load4(rsrc_in, vo, 0);
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const unsigned sa = s_wbase[i] + (unsigned)s_kc * 8192u;
#pragma unroll
            for (int q = 0; q < 8; ++q) a_nxt[i][q] = bk_load4(rsrc_w, lane16, sa + q * 1024u);
        }
    };
    auto lstore = [&](int buf) {
        float* Bb = Bs + buf * BK_BN * BK_LDK;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const float x[4] = {bk_act<ACT>(rb[rr].x, a.slope), bk_act<ACT>(rb[rr].y, a.slope), bk_act<ACT>(rb[rr].z, a.slope), bk_act<ACT>(rb[rr].w, a.slope)};
            f16x4b hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const _Float16 h = (_Float16)x[e];
                hi[e] = h;
                lo[e] = (_Float16)((x[e] - (float)h) * kBkLoScale);
            }
            unsigned char* d = reinterpret_cast<unsigned char*>(Bb + (srow + CPR * rr) * BK_LDK) + 8 * quad;
            *reinterpret_cast<f16x4b*>(d) = hi;
            *reinterpret_cast<f16x4b*>(d + 2 * BK_KC) = lo;
        }
    };

    f32x16 am[2][2], ac[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { am[i][j][e] = 0.f; ac[i][j][e] = 0.f; }

    // ---- prologue: first chunk into LDS buffer 0 ----
    int tile = (int)(u0 / bk.nchunks);
    int kc = (int)(u0 - (long long)tile * bk.nchunks);
    int seg_start_kc = kc;
    const int n_units = (int)(u1 - u0);
    stage_tile(tile, kc);
    int cur_g = s_g, cur_mt = s_mt, cur_nt = s_nt;
    gload();
    lstore(0);
    float4 a_cur[2][8];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int q = 0; q < 8; ++q) a_cur[i][q] = a_nxt[i][q];
    __syncthreads();
    bool bad = false;

    int cur = 0;
    for (int it = 0; it < n_units; ++it) {
        const bool has_next = (it + 1 < n_units);
        if (has_next) {
            stage_advance();
            gload();
        }
        // -- 48 MFMAs on the current chunk --
        const unsigned char* Bh = reinterpret_cast<const unsigned char*>(Bs + cur * BK_BN * BK_LDK + (wn * 64 + l31) * BK_LDK) + 16 * lh;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            union { float4 f; f16x8b h; } ah[2], al[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) { ah[i].f = a_cur[i][2 * st]; al[i].f = a_cur[i][2 * st + 1]; }
            f16x8b bh[2], bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const f16x8b*>(Bh + j * 32 * BK_LDK * 4 + 32 * st);
                bl[j] = *reinterpret_cast<const f16x8b*>(Bh + j * 32 * BK_LDK * 4 + 32 * st + 2 * BK_KC);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h, bh[j], am[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i].h, bl[j], ac[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i].h, bh[j], ac[i][j], 0, 0, 0);
        }
        // -- stage the next chunk --
        if (has_next) lstore(cur ^ 1);
        // -- end of this tile's segment? --
        if (kc == bk.nchunks - 1 || !has_next) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { am[i][j][e] = fmaf(ac[i][j][e], kBkLoInv, am[i][j][e]); ac[i][j][e] = 0.f; }
            const bool seg_first = (seg_start_kc == 0), seg_last = (kc == bk.nchunks - 1);
            if (!seg_first) {
                // head of this range: the tile belongs to the workgroup holding its first chunk.  Publish the raw partial sums:
                // write-through (sc1) 16-byte stores (one contiguous KiB per wave instruction), every wave drains, barrier, one
                // relaxed agent-scope flag (guide G16 R1)
                const __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc(bk.ws, 0, bk.ws_bytes, 0x00020000);
                const unsigned wbase = (unsigned)r * (unsigned)(256 * 256) + (unsigned)tid * 16u;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            u32x4b v;
                            v.x = __float_as_uint(am[i][j][4 * e4]); v.y = __float_as_uint(am[i][j][4 * e4 + 1]);
                            v.z = __float_as_uint(am[i][j][4 * e4 + 2]); v.w = __float_as_uint(am[i][j][4 * e4 + 3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_ws, wbase + (unsigned)(((i * 2 + j) * 4 + e4) * (256 * 16)), 0, 16 /* sc1 */);
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(bk.flags + r, bk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (!seg_last) {
                    // owner of a tile that continues in the following range(s): add their partials in range order (deterministic)
                    const long long t1 = ((long long)tile + 1) * bk.nchunks;
                    int rr_end = r + 1;
                    while (rr_end < bk.G && bk_u0(rr_end, bk) < t1) ++rr_end;
                    if (tid == 0) {
                        for (int rr = r + 1; rr < rr_end; ++rr) {
                            if (bk_u0(rr + 1, bk) <= bk_u0(rr, bk)) continue;
                            unsigned spins = 0;
                            while (__hip_atomic_load(bk.flags + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != bk.epoch) {
                                __builtin_amdgcn_s_sleep(2);
                                if (++spins > (1u << 20)) { atomicOr(bk.err, 2); break; }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    for (int rr = r + 1; rr < rr_end; ++rr) {
                        if (bk_u0(rr + 1, bk) <= bk_u0(rr, bk)) continue;
                        const float* wsp = bk.ws + (size_t)rr * (256 * 64) + (size_t)tid * 4;
                        float4 pv[16];
#pragma unroll
                        for (int pc = 0; pc < 16; ++pc) pv[pc] = *reinterpret_cast<const float4*>(wsp + (size_t)pc * (256 * 4));
#pragma unroll
                        for (int i = 0; i < 2; ++i)
#pragma unroll
                            for (int j = 0; j < 2; ++j)
#pragma unroll
                                for (int e4 = 0; e4 < 4; ++e4) {
                                    const float4 v = pv[(i * 2 + j) * 4 + e4];
                                    am[i][j][4 * e4] += v.x; am[i][j][4 * e4 + 1] += v.y; am[i][j][4 * e4 + 2] += v.z; am[i][j][4 * e4 + 3] += v.w;
                                }
                    }
                }
#pragma unroll
                for (int i = 0; i < 2; ++i)
                    bk_epilogue(a, am[i], cur_g, cur_mt * BK_BM + (2 * wm + i) * 32, cur_nt * BK_BN + wn * 64, lane, bad);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) am[i][j][e] = 0.f;
            seg_start_kc = 0;
            if (has_next) tile_coords(tile + 1, cur_g, cur_mt, cur_nt);
        }
        __syncthreads();
        cur ^= 1;
#pragma unroll
        for (int i = 0; i < 2; ++i)
#pragma unroll
            for (int q = 0; q < 8; ++q) a_cur[i][q] = a_nxt[i][q];
        if (++kc == bk.nchunks) { kc = 0; ++tile; }
    }
    if (bad) atomicOr(bk.err, 8);
}

int g_bk_enable = -1;       // ADK_CONV_BK16=1: where the heuristic says so; 2: wherever the shape allows (tuning).  Default 0: measured no
                            // faster than the 64-wide kernel (see below)
int g_bk_min_tiles = 48;    // ADK_BK16_MIN_TILES
int g_bk_split = 4;         // ADK_BK16_SPLIT: most workgroups sharing one tile
}  // namespace

bool conv_bk16_pick(const ConvArgs& a, bool force) {
    if (g_bk_enable < 0) {
        const char* e = getenv("ADK_CONV_BK16"); g_bk_enable = e ? atoi(e) : 0;
        e = getenv("ADK_BK16_MIN_TILES"); if (e && atoi(e) > 0) g_bk_min_tiles = atoi(e);
        e = getenv("ADK_BK16_SPLIT"); if (e && atoi(e) > 0) g_bk_split = atoi(e);
    }
    if ((!g_bk_enable && !force) || !conv_mfma_supported(a)) return false;
    if (a.cout_g < 128) return false;                                  // narrow layers keep the 64-row tiles
    if (force || g_bk_enable >= 2) return true;
    const long long tiles = (long long)((a.cout_g + BK_BM - 1) / BK_BM) * ((a.n_total + BK_BN - 1) / BK_BN) * a.groups;
    return tiles >= g_bk_min_tiles && a.ktot >= 512;
}

int launch_conv_bk16(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    if (a.n_total == 0) return ADK_OK;
    (void)conv_mfma_workspace_bytes(nullptr);
    (void)conv_bk16_pick(a, true);                                     // reads the env knobs
    constexpr size_t lds = 2ull * BK_BN * BK_LDK * sizeof(float);
    BkArgs bk;
    bk.m_tiles = (a.cout_g + BK_BM - 1) / BK_BM;
    bk.n_tiles = (a.n_total + BK_BN - 1) / BK_BN;
    bk.nchunks = (a.ktot + BK_KC - 1) / BK_KC;
    bk.cpt = a.cin_g / 32;
    bk.kgroups = (a.ktot + BK_KC - 1) / BK_KC * (BK_KC / 8);
    bk.mt32_per_g = (a.cout_g + 31) / 32;
    bk.inv_t_out = 1.0f / (float)a.t_out;
    {
        const unsigned long long inb = (unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull;
        const unsigned long long wb = (unsigned long long)a.groups * bk.mt32_per_g * bk.kgroups * 1024ull;
        if (inb >= 0x80000000ull || wb >= 0xfff00000ull || a.n_total >= (1 << 24))
            return fail(ADK_ERR_SHAPE, "conv: problem too large for the 32-bit buffer addressing of the MFMA kernel");
        bk.in_bytes = (unsigned)inb; bk.w_bytes = (unsigned)wb;
    }
    const long long tiles = (long long)bk.m_tiles * bk.n_tiles * a.groups;
    bk.total = tiles * bk.nchunks;
    long long G = ws.workgroups > 0 ? std::min<long long>(ws.workgroups, 256) : 256;       // one workgroup per CU
    const long long by_units = (bk.total + 1) / 2;
    if (G > by_units) G = (by_units + 7) / 8 * 8;
    if (G > tiles * g_bk_split) G = (tiles * g_bk_split + 7) / 8 * 8;
    if (G > 256) G = 256;
    bk.G = (int)G;
    const size_t part_bytes = (size_t)bk.G * 256 * 64 * sizeof(float);
    if (!ws.ptr || part_bytes > ws.flags_offset || ws.flags_offset + (size_t)bk.G * sizeof(unsigned) > ws.bytes)
        return fail(ADK_ERR_STATE, "conv: stream-K workspace missing or too small");
    bk.ws = ws.ptr;
    bk.ws_bytes = (unsigned)part_bytes;
    bk.flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws.ptr) + ws.flags_offset);
    bk.epoch = ++ws.epoch;
    if (bk.epoch == 0) bk.epoch = ++ws.epoch;
    bk.err = flags_word();
    {
        static bool attr_set_dev[kMaxDevices] = {};
        bool& attr_set = attr_set_dev[current_device()];
        if (!attr_set) {
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bk16_kernel<ADK_ACT_ELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bk16_kernel<ADK_ACT_LEAKY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bk16_kernel<ADK_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    if (a.act_in == ADK_ACT_ELU) hipLaunchKernelGGL(conv_bk16_kernel<ADK_ACT_ELU>, dim3(bk.G), dim3(256), lds, s, a, bk);
    else if (a.act_in == ADK_ACT_LEAKY) hipLaunchKernelGGL(conv_bk16_kernel<ADK_ACT_LEAKY>, dim3(bk.G), dim3(256), lds, s, a, bk);
    else if (a.act_in == ADK_ACT_NONE) hipLaunchKernelGGL(conv_bk16_kernel<ADK_ACT_NONE>, dim3(bk.G), dim3(256), lds, s, a, bk);
    else return fail(ADK_ERR_ARG, "conv: unsupported input activation for the big-tile kernel");
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

}  // namespace adk
