// Big-tile split-f16 implicit-GEMM conv for the DEEP layers (few output steps per stream, long K): the grouped K11 convs of
// vocoder stages 0-1, the K7 convs of encoder blocks 2-3, the strided / transposed convs and the wide 1x1s.  Same arithmetic and
// the same persistent stream-K schedule as conv_sk_kernel<SPLIT> (conv_mfma.hip) -- replaces F.conv1d / F.conv_transpose1d of
// CausalConv1d.inference / CausalConvTranspose1d.inference (layers/conv_layer.py:153-156, 194-197) fused with input
// activation, bias and residual add -- but built around what the first-round profile of that kernel showed
// (profiles/r1_pmc_split16_*): 64x64 workgroup tiles re-gathered every activation row once per m-tile and re-streamed the
// weight panel once per n-tile (4x the compulsory traffic), and with one 64-deep chunk of loads in flight per workgroup every
// iteration waited out a full L2 round trip (matrix cores 13 % busy).  Here:
//   * wave tile 64 x 64 (2 x 2 MFMA tiles, each A / B fragment feeds two MFMA triples), workgroup tile 256 x 128 or 128 x 256
//     (8 waves): every activation row of a group with <= 256 channels is gathered ONCE per launch, the weight panel is
//     streamed once per 128 / 256 columns;
//   * both operands reach LDS by LDS-DMA (global_load_lds_dwordx4, no staging registers) into a 3-stage ring of 32-deep
//     K slices: two slices (96 KB) are in flight behind the one being multiplied, one raw s_barrier per slice, counted vmcnt;
//   * the activations are staged RAW (f32): the DMA cannot convert, so the input activation and the f16 hi / lo split are
//     applied when a wave reads its B fragment (once per 64 output channels; the f16 matrix cores leave the VALU idle);
//     LDS rows are 128 bytes, XOR-swizzled on the SOURCE address so that the fragment reads are bank-conflict free;
//   * weights come in the adk_pack_weights_split16 fragment order, so their DMA is a straight lane-linear copy.
//
// MEASURED (round 2, tools/kbench, 256 streams, profiles/r2_gk16_experiment.md): correct (max |d| 4e-6 against the exact-f32
// kernel, parity tests green end to end) and SLOWER than conv_sk16 on every layer -- grouped K11 256-ch 58 vs 51 us, 128-ch
// 55-65 vs 52 us, encoder K7 47 vs 28 us.  Two reasons, both visible in the numbers: (1) the LDS-DMA path delivered ~12 GB/s per
// CU here (3 TB/s chip-wide; MI355X_MICROARCH.md quotes 25 GB/s per loader wave) where the 64-wide kernel pulls ~27 GB/s per CU
// through plain 16-byte loads, so a 32-deep slice takes 2.5-4 us instead of the 0.75 us the matrix cores need; (2) converting at
// fragment-read time costs ~400 VALU instructions per slice per wave, twice the MFMA issue time.  With few tiles the serial
// reduction of 64-128 KB partial slabs by the tile owner (22 GB/s) made it 3-6x slower still (first version: 141 vs 29 us).
// It therefore stays OPT-IN (ADK_CONV_GK16=1 / ADK_IMPL_SPLIT16_GK) as the record of that experiment and a test bed.
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

namespace adk {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8g __attribute__((ext_vector_type(8)));
typedef unsigned u32x4g __attribute__((ext_vector_type(4)));

namespace {
constexpr float kGkLoScale = 2048.f, kGkLoInv = 1.f / 2048.f;
#ifndef ADK_GK16_DBG
#define ADK_GK16_DBG 0      // tuning experiments only: 1 = fragments used as read (no activation, no hi / lo conversion; results wrong)
#endif
constexpr int GK_KS = 32;          // K slice per stage: one tap x 32 channels
constexpr int GK_NST = 3;          // LDS ring depth

typedef const void __attribute__((address_space(1)))* gk_gptr;
typedef void __attribute__((address_space(3)))* gk_lptr;

struct GkArgs {
    float* ws;            // partial-tile workspace: [G][threads][64] floats
    unsigned* flags;      // [G] publish flags (epoch-tagged)
    unsigned epoch;
    int G;
    int m_tiles, n_tiles, nstages;      // workgroup tiles per group, K slices per tile
    int cpt;              // 32-channel blocks per tap
    int ksteps16;         // 16-k chunks per 32-row m-tile in the packed weights (K padded to a multiple of 64)
    int mt32_per_g;
    unsigned ws_bytes;
    int* err;
    float inv_t_out;
    long long total;      // tiles * nstages
};

template <int ACT>
__device__ __forceinline__ float gk_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

__device__ __forceinline__ long long gk_u0(int r, const GkArgs& gk) { return (long long)r * gk.total / gk.G; }

__device__ __forceinline__ int gk_div(int n, int d, float inv_d) {      // n / d for 0 <= n < 2^24
    int q = (int)(__int2float_rn(n) * inv_d);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; }
    return q;
}

// bias, residual, output activation, store for one 32 x 64 accumulator block (two n-tiles) of a wave
__device__ __forceinline__ void gk_epilogue(const ConvArgs& a, const f32x16 (&acc)[2], int g, int ml0, int n0w, int lane, bool& bad) {
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

// WM x WN waves, each a 64 x 64 output block.  One iteration = one 32-deep K slice of one tile.
template <int WM, int WN, int ACT>
__global__ __launch_bounds__(64 * WM * WN, (WM * WN) >= 8 ? 2 : 1) void conv_gk16_kernel(ConvArgs a, GkArgs gk) {
    constexpr int NW = WM * WN, NT = 64 * NW;
    constexpr int BM = 64 * WM, BN = 64 * WN;
    constexpr int XB = BN * 128;                      // bytes of the activation part of a stage: BN columns x 32 floats
    constexpr int WB = (BM / 32) * 4096;              // bytes of the weight part: per 32-row m-tile 2 chunks x (hi | lo) x 1 KiB
    constexpr int STB = XB + WB;
    constexpr int NXI = BN / 8;                       // DMA wave-instructions per stage: 8 columns each ...
    constexpr int NWI = (BM / 32) * 4;                // ... and 1 KiB of weight fragments each
    static_assert(NXI % NW == 0 && NWI % NW == 0, "stage DMA must divide evenly over the waves");
    constexpr int XPW = NXI / NW, WPW = NWI / NW;     // per wave
    constexpr int PPW = XPW + WPW;                    // DMA instructions a wave issues per stage
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];     // [GK_NST][STB]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-contiguous range of work units (block b runs on XCD b % 8: speed only, any placement is correct)
    const int r = (int)(blockIdx.x & 7) * (gk.G >> 3) + (int)(blockIdx.x >> 3);
    const long long u0 = gk_u0(r, gk), u1 = gk_u0(r + 1, gk);
    if (u0 >= u1) return;

    const unsigned row_bytes = (unsigned)a.in_ch * 4u;
    const unsigned ring_bytes = (unsigned)a.in_rows * row_bytes;
    const unsigned dil_bytes = (unsigned)a.dilation * row_bytes;
    const unsigned char* in_bytes = reinterpret_cast<const unsigned char*>(a.in);
    const unsigned char* w_bytes = reinterpret_cast<const unsigned char*>(a.wfrag);

    // ---- DMA state: describes the NEXT stage to be issued ----
    int s_tile, s_st;                                  // wave-uniform: tile and K slice
    int s_g = 0, s_mt = 0, s_nt = 0;
    int s_tap, s_cblk;                                 // tap / 32-channel block of that slice
    unsigned colb[XPW], rowb[XPW];                     // this lane's column: stream + channel + 16-byte piece offset; ring row offset (tap applied)
    unsigned wofs[WPW];                                // this wave's weight pieces: byte offset of slice 0 (0xffffffff: m-tile beyond the group)

    auto tile_coords = [&](int tile, int& g, int& mt, int& nt) {
        mt = tile % gk.m_tiles;
        const int rest = tile / gk.m_tiles;
        nt = rest % gk.n_tiles;
        g = rest / gk.n_tiles;
    };
    auto stage_tile = [&](int tile, int st0) {
        tile_coords(tile, s_g, s_mt, s_nt);
        s_tile = tile; s_st = st0;
        s_tap = st0 / gk.cpt; s_cblk = st0 - s_tap * gk.cpt;
        const unsigned tap_bytes = (unsigned)s_tap * dil_bytes;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int xi = wave + NW * i;                         // this wave's i-th column octet of the tile
            const int nl = 8 * xi + (lane >> 3);                  // column inside the tile
            int n = s_nt * BN + nl;
            if (n >= a.n_total) n = a.n_total - 1;                // padded columns re-read the last valid one (never stored)
            const int b = gk_div(n, a.t_out, gk.inv_t_out), t = n - b * a.t_out;
            int row = a.in_row0 + t * a.stride;
            if (row >= a.in_rows) row -= a.in_rows;
            unsigned rbv = (unsigned)row * row_bytes + tap_bytes;
            if (rbv >= ring_bytes) rbv -= ring_bytes;
            rowb[i] = rbv;
            const unsigned q = (unsigned)(lane & 7) ^ (unsigned)((nl >> 1) & 7);      // logical 16-byte piece this lane fetches
            colb[i] = (unsigned)b * ring_bytes + (unsigned)(a.in_choff + s_g * a.in_gstride) * 4u + q * 16u;
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int wi = wave + NW * i;
            const int mtl = wi >> 2, piece = wi & 3;              // m-tile inside the workgroup tile, 1 KiB piece of its slice
            const int mt32 = s_mt * (BM / 32) + mtl;
            wofs[i] = mt32 < gk.mt32_per_g ? (unsigned)((s_g * gk.mt32_per_g + mt32) * gk.ksteps16) * 2048u + (unsigned)piece * 1024u + (unsigned)lane * 16u
                                           : 0xffffffffu;
        }
    };
    auto stage_advance = [&]() {
        ++s_st;
        if (s_st == gk.nstages) { stage_tile(s_tile + 1, 0); return; }
        if (++s_cblk == gk.cpt) {
            s_cblk = 0; ++s_tap;
#pragma unroll
            for (int i = 0; i < XPW; ++i) {
                unsigned rbv = rowb[i] + dil_bytes;
                if (rbv >= ring_bytes) rbv -= ring_bytes;
                rowb[i] = rbv;
            }
        }
    };
    auto issue = [&](int buf) {                        // PPW LDS-DMA instructions: the staged slice -> ring buffer `buf`
        unsigned char* st = lds + buf * STB;
        const unsigned cb = (unsigned)s_cblk * 128u;
#pragma unroll
        for (int i = 0; i < XPW; ++i) {
            const int xi = wave + NW * i;
            __builtin_amdgcn_global_load_lds((gk_gptr)(in_bytes + (size_t)colb[i] + rowb[i] + cb), (gk_lptr)(st + xi * 1024), 16, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < WPW; ++i) {
            const int wi = wave + NW * i;
            const unsigned o = wofs[i] != 0xffffffffu ? wofs[i] + (unsigned)s_st * 4096u : (unsigned)lane * 16u;   // beyond the group: any valid bytes
            __builtin_amdgcn_global_load_lds((gk_gptr)(w_bytes + o), (gk_lptr)(st + XB + wi * 1024), 16, 0, 0);
        }
    };

    f32x16 am[2][2], ac[2][2];                         // main / cross-term accumulators of the 2 x 2 MFMA tiles
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { am[i][j][e] = 0.f; ac[i][j][e] = 0.f; }

    int tile = (int)(u0 / gk.nstages);
    int st = (int)(u0 - (long long)tile * gk.nstages);
    int seg_start = st;
    const int n_units = (int)(u1 - u0);
    stage_tile(tile, st);
    int cur_g = s_g, cur_mt = s_mt, cur_nt = s_nt;
    // prologue: GK_NST - 1 slices in flight
#pragma unroll
    for (int p = 0; p < GK_NST - 1; ++p)
        if (p < n_units) { if (p > 0) stage_advance(); issue(p); }
    int issued = n_units < GK_NST - 1 ? n_units : GK_NST - 1;
    // LDS addresses of this lane's fragments inside a stage
    unsigned xoff[2][2][2];                            // [n-tile j][k-step][piece]: swizzled byte offsets of the B fragment halves
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int nl = wn * 64 + j * 32 + l31;
        const unsigned sw = (unsigned)((nl >> 1) & 7);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks)
#pragma unroll
            for (int pc = 0; pc < 2; ++pc)
                xoff[j][ks][pc] = (unsigned)nl * 128u + ((((unsigned)(4 * ks + 2 * lh + pc)) ^ sw) * 16u);
    }
    const unsigned aoff = (unsigned)XB + (unsigned)(2 * wm) * 4096u + (unsigned)lane * 16u;
    bool bad = false;

    int buf = 0;
    for (int it = 0; it < n_units; ++it) {
        // the slice of this iteration has landed (this wave's share: all but the newest issued - it - 1 slices) ...
        if (issued - it - 1 >= GK_NST - 2 && GK_NST >= 3) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((GK_NST - 2) * PPW) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");        // ... every wave's share has, and every wave is done with the previous slice
        if (issued < n_units) {                        // refill the buffer the previous iteration read
            stage_advance();
            issue((buf + GK_NST - 1) % GK_NST);
            ++issued;
        }
        // Fragment reads go through inline asm: hipcc cannot tell the LDS-DMA in flight (into the OTHER ring buffers) from these
        // reads and would wait vmcnt(0) in front of its own ds_read, draining the prefetch every iteration.  asm reads are not
        // counted by the compiler: the lgkmcnt waits below name their destinations ("+v"), so nothing uses a register before
        // its data has landed (guide 5.7, form ii).
        const unsigned sbase = (unsigned)(buf * STB);
        u32x4g ra[2][4];                                   // [k-step][m-tile i: hi, lo]: A fragments as read
        u32x4g rx[2][4];                                   // [k-step][n-tile j: piece 0, 1]: raw f32 activations
        auto read_frags = [&](int ks) {
            const unsigned aa = sbase + aoff;
            if (ks == 0) {
                asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:4096\n\tds_read_b128 %3, %4 offset:5120"
                             : "=&v"(ra[0][0]), "=&v"(ra[0][1]), "=&v"(ra[0][2]), "=&v"(ra[0][3]) : "v"(aa) : "memory");
            } else {
                asm volatile("ds_read_b128 %0, %4 offset:2048\n\tds_read_b128 %1, %4 offset:3072\n\tds_read_b128 %2, %4 offset:6144\n\tds_read_b128 %3, %4 offset:7168"
                             : "=&v"(ra[1][0]), "=&v"(ra[1][1]), "=&v"(ra[1][2]), "=&v"(ra[1][3]) : "v"(aa) : "memory");
            }
            const unsigned x00 = sbase + xoff[0][ks][0], x01 = sbase + xoff[0][ks][1], x10 = sbase + xoff[1][ks][0], x11 = sbase + xoff[1][ks][1];
            asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %5\n\tds_read_b128 %2, %6\n\tds_read_b128 %3, %7"
                         : "=&v"(rx[ks][0]), "=&v"(rx[ks][1]), "=&v"(rx[ks][2]), "=&v"(rx[ks][3]) : "v"(x00), "v"(x01), "v"(x10), "v"(x11) : "memory");
        };
        auto wait_frags = [&](int ks) {                    // everything read so far has landed (reads return in order)
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(ra[ks][0]), "+v"(ra[ks][1]), "+v"(ra[ks][2]), "+v"(ra[ks][3]),
                                                  "+v"(rx[ks][0]), "+v"(rx[ks][1]), "+v"(rx[ks][2]), "+v"(rx[ks][3]) :: "memory");
        };
        auto as_h = [](const u32x4g& v) { union { u32x4g u; f16x8g h; } c; c.u = v; return c.h; };
        read_frags(0);
#pragma unroll
        for (int ks = 0; ks < 2; ++ks) {
            wait_frags(ks);
            if (ks == 0) read_frags(1);                    // in flight under the conversion and the MFMAs of k-step 0
            f16x8g Bh[2], Bl[2];
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                const u32x4g u = rx[ks][2 * j], v = rx[ks][2 * j + 1];
#if ADK_GK16_DBG & 1
                Bh[j] = as_h(u); Bl[j] = as_h(v);      // what a pre-split ring (hi | lo f16 planes, same bytes) would cost
                continue;
#endif
                const float x[8] = {__uint_as_float(u.x), __uint_as_float(u.y), __uint_as_float(u.z), __uint_as_float(u.w),
                                    __uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w)};
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const float y = gk_act<ACT>(x[e], a.slope);
                    const _Float16 h = (_Float16)y;
                    Bh[j][e] = h;
                    Bl[j][e] = (_Float16)((y - (float)h) * kGkLoScale);
                }
            }
            const f16x8g Ah[2] = {as_h(ra[ks][0]), as_h(ra[ks][2])}, Al[2] = {as_h(ra[ks][1]), as_h(ra[ks][3])};
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) am[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i], Bh[j], am[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[i], Bl[j], ac[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) ac[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[i], Bh[j], ac[i][j], 0, 0, 0);
        }
        // ---- end of this tile's segment? ----
        const bool last_it = (it + 1 == n_units);
        if (st == gk.nstages - 1 || last_it) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { am[i][j][e] = fmaf(ac[i][j][e], kGkLoInv, am[i][j][e]); ac[i][j][e] = 0.f; }
            const bool seg_first = (seg_start == 0), seg_last = (st == gk.nstages - 1);
            if (!seg_first) {
                // head of this range: the tile belongs to the workgroup holding its first slice.  Publish the raw partial sums:
                // write-through (sc1) 16-byte stores, every wave drains, barrier, one relaxed agent-scope flag (guide G16 R1)
                const __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc(gk.ws, 0, gk.ws_bytes, 0x00020000);
                // slab layout [range][16 pieces][thread][16 B]: every store / load instruction of a wave is one contiguous KiB
                const unsigned wbase = (unsigned)r * (unsigned)(NT * 256) + (unsigned)tid * 16u;
#pragma unroll
                for (int i = 0; i < 2; ++i)
#pragma unroll
                    for (int j = 0; j < 2; ++j)
#pragma unroll
                        for (int e4 = 0; e4 < 4; ++e4) {
                            u32x4g v;
                            v.x = __float_as_uint(am[i][j][4 * e4]); v.y = __float_as_uint(am[i][j][4 * e4 + 1]);
                            v.z = __float_as_uint(am[i][j][4 * e4 + 2]); v.w = __float_as_uint(am[i][j][4 * e4 + 3]);
                            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_ws, wbase + (unsigned)(((i * 2 + j) * 4 + e4) * (NT * 16)), 0, 16 /* sc1 */);
                        }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __syncthreads();
                if (tid == 0) __hip_atomic_store(gk.flags + r, gk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (!seg_last) {
                    // owner of a tile that continues in the following range(s): add their partials in range order (deterministic)
                    const long long t1 = ((long long)tile + 1) * gk.nstages;
                    int rr_end = r + 1;
                    while (rr_end < gk.G && gk_u0(rr_end, gk) < t1) ++rr_end;
                    if (tid == 0) {
                        for (int rr = r + 1; rr < rr_end; ++rr) {
                            if (gk_u0(rr + 1, gk) <= gk_u0(rr, gk)) continue;
                            unsigned spins = 0;
                            while (__hip_atomic_load(gk.flags + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != gk.epoch) {
                                __builtin_amdgcn_s_sleep(2);
                                if (++spins > (1u << 20)) { atomicOr(gk.err, 2); break; }
                            }
                        }
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                    }
                    __syncthreads();
                    for (int rr = r + 1; rr < rr_end; ++rr) {
                        if (gk_u0(rr + 1, gk) <= gk_u0(rr, gk)) continue;
                        const float* wsp = gk.ws + (size_t)rr * (NT * 64) + (size_t)tid * 4;
                        float4 pv[16];                             // all 16 pieces of this contributor in flight, then added in order
#pragma unroll
                        for (int pc = 0; pc < 16; ++pc) pv[pc] = *reinterpret_cast<const float4*>(wsp + (size_t)pc * (NT * 4));
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
                    gk_epilogue(a, am[i], cur_g, cur_mt * BM + (2 * wm + i) * 32, cur_nt * BN + wn * 64, lane, bad);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) am[i][j][e] = 0.f;
            seg_start = 0;
            if (!last_it) tile_coords(tile + 1, cur_g, cur_mt, cur_nt);
        }
        buf = (buf + 1) % GK_NST;
        if (++st == gk.nstages) { st = 0; ++tile; }
    }
    if (bad) atomicOr(gk.err, 8);
}

int g_gk_enable = -1;       // ADK_CONV_GK16=1: AUTO takes this kernel where its heuristic says so; 2|3|4: force 256x128 | 128x256 | 128x128.
                            // Default 0: measured slower than the 64-wide stream-K kernel on every layer of the path (see the header)

template <int WM, int WN>
int launch_gk(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    constexpr int BM = 64 * WM, BN = 64 * WN, NT = 64 * WM * WN;
    constexpr size_t lds = (size_t)GK_NST * (BN * 128 + (BM / 32) * 4096);
    GkArgs gk;
    gk.m_tiles = (a.cout_g + BM - 1) / BM;
    gk.n_tiles = (a.n_total + BN - 1) / BN;
    gk.nstages = a.ktot / GK_KS;
    gk.cpt = a.cin_g / 32;
    gk.ksteps16 = (a.ktot + 63) / 64 * 4;
    gk.mt32_per_g = (a.cout_g + 31) / 32;
    gk.inv_t_out = 1.0f / (float)a.t_out;
    const long long tiles = (long long)gk.m_tiles * gk.n_tiles * a.groups;
    gk.total = tiles * gk.nstages;
    // one persistent workgroup per CU (the LDS ring fills a CU), fewer for small problems: at least 2 slices each
    long long G = ws.workgroups > 0 ? std::min<long long>(ws.workgroups, 256) : 256;
    const long long by_units = (gk.total + 1) / 2;
    if (G > by_units) G = (by_units + 7) / 8 * 8;
    static int max_split = -1;                                   // ADK_GK16_SPLIT: most workgroups sharing one tile (default 4)
    if (max_split < 0) { const char* e = getenv("ADK_GK16_SPLIT"); max_split = (e && atoi(e) > 0) ? atoi(e) : 4; }
    if (G > tiles * max_split) G = (tiles * max_split + 7) / 8 * 8;
    if (G > 256) G = 256;
    gk.G = (int)G;
    const size_t part_bytes = (size_t)gk.G * NT * 64 * sizeof(float);
    if (!ws.ptr || part_bytes > ws.flags_offset || ws.flags_offset + (size_t)gk.G * sizeof(unsigned) > ws.bytes)
        return fail(ADK_ERR_STATE, "conv: stream-K workspace missing or too small");
    gk.ws = ws.ptr;
    gk.ws_bytes = (unsigned)part_bytes;
    gk.flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws.ptr) + ws.flags_offset);
    gk.epoch = ++ws.epoch;
    if (gk.epoch == 0) gk.epoch = ++ws.epoch;
    gk.err = flags_word();
    {
        static bool attr_set_dev[kMaxDevices] = {};              // per (WM, WN) instantiation and device
        bool& attr_set = attr_set_dev[current_device()];
        if (!attr_set) {
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gk16_kernel<WM, WN, ADK_ACT_ELU>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gk16_kernel<WM, WN, ADK_ACT_LEAKY>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gk16_kernel<WM, WN, ADK_ACT_NONE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    auto go = [&](auto kern) -> int {
        hipLaunchKernelGGL(kern, dim3(gk.G), dim3(NT), lds, s, a, gk);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    if (a.act_in == ADK_ACT_ELU) return go(conv_gk16_kernel<WM, WN, ADK_ACT_ELU>);
    if (a.act_in == ADK_ACT_LEAKY) return go(conv_gk16_kernel<WM, WN, ADK_ACT_LEAKY>);
    if (a.act_in == ADK_ACT_NONE) return go(conv_gk16_kernel<WM, WN, ADK_ACT_NONE>);
    return fail(ADK_ERR_ARG, "conv: unsupported input activation for the big-tile kernel");
}
}  // namespace

// 0: not taken; 1: 256 x 128 tiles; 2: 128 x 256 tiles; 3: 128 x 128 tiles (4 waves; the default shape)
int conv_gk16_pick(const ConvArgs& a, bool force) {
    if (g_gk_enable < 0) { const char* e = getenv("ADK_CONV_GK16"); g_gk_enable = e ? atoi(e) : 0; }
    if ((!g_gk_enable && !force) || !conv_mfma_supported(a)) return 0;
    if ((unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull >= 0xf0000000ull) return 0;      // 32-bit byte offsets inside the arena view
    if (a.cout_g < 128) return 0;                                     // narrow layers keep the 64-row tiles
    const int forced = g_gk_enable >= 2 ? g_gk_enable - 1 : 0;       // ADK_CONV_GK16=2|3|4: force a shape (tuning)
    const int shape = forced ? forced : 3;
    const long long bm = shape == 1 ? 256 : 128, bn = shape == 2 ? 256 : 128;
    const long long tiles = ((a.cout_g + bm - 1) / bm) * ((a.n_total + bn - 1) / bn) * a.groups;
    // The partial sums of a tile cut by a range boundary travel through memory (64 KB per cut of a 128 x 128 tile) and the
    // owner adds them one contributor after the other: big tiles pay when there are enough of them that a tile is shared by
    // few workgroups (measured: a launch with 10-30 tiles over 256 workgroups spends most of its time in that reduction) and
    // K is long enough to amortise the prologue.  Everything else stays on the 64-wide tiles.
    if (!forced && !force && (tiles < 48 || a.ktot < 256)) return 0;
    return shape;
}

int launch_conv_gk16(const ConvArgs& a, hipStream_t s, Workspace& ws, bool force) {
    if (a.n_total == 0) return ADK_OK;
    (void)conv_mfma_workspace_bytes(nullptr);
    const int shape = conv_gk16_pick(a, force);
    if (!shape) return fail(ADK_ERR_SHAPE, "conv: the big-tile kernel needs cin_g % 32 == 0, >= 128 output channels per group, 16-byte aligned rows");
    if (shape == 1) return launch_gk<4, 2>(a, s, ws);
    if (shape == 2) return launch_gk<2, 4>(a, s, ws);
    return launch_gk<2, 2>(a, s, ws);
}

}  // namespace adk
