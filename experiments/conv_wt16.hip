// NOT COMPILED, NOT PART OF THE LIBRARY.  Round 6's wide-tile kernel for the six grouped convs of the vocoder's first stage: built, hooked into
// launch_conv_sk16 (`conv_wt16_preferred`), tested (tests/test_gpu_b256.py::test_wide_tile_kernel_agrees_with_the_stream_k_kernel of that commit:
// five stream counts, both tile heights, 1 ... 8 K parts, every conv it supports: within 2e-5 of the stream-K run, two runs bit-identical; the
// benched configuration against the CPU oracle), measured, and taken out again: alone it matches the stream-K launches at best (27.6 - 30.6 us inside
// the kernel against ~30), inside the three-stream pipeline it LOSES 3.7 - 9.6 % (profiles/r6_wt16.md has the phase clocks and the A/B).
// It lived in audiodec_amd/csrc/ next to conv_mfma.hip and uses adk_common.h of that commit (+ the five declarations it added there);
// experiments/wt16_bench.py and wt16_trace.py are its measurement tools (they use the options `wt16`, `wt16_rows`, `wt16_buffers` this file defines).
//
// conv_wt16 -- wide tiles for the wide grouped convs of the vocoder's first stage (HiFiGANResidualBlock.inference,
// models/vocoder/modules/residual_block.py:99-105, at 3 x 256 channels: K 11, 5 steps per stream and frame), round 6.
//
// The stream-K kernel (conv_mfma.hip) gives every wave a 32 x 32 block of a 64 x 64 tile: per 64-deep chunk a workgroup requests 32 KiB of
// operands for 48 MFMAs -- 683 bytes per MFMA, 338 MB per conv at 256 streams, and a CU takes in ~25 bytes per clock: the launch sits on the
// cache-to-CU delivery at ~19 % of the matrix cores whatever its loop does (profiles/r2_sk16_analysis.md).  Fewer bytes per
// MFMA need wider tiles, and wider tiles need both operands in LDS without a detour through registers.  With a SHADOW ring as input
// (adk_op_desc.in_shadow: the split-f16 operand form of act(x) is in memory, written once by the producer's epilogue) that is a plain copy:
//   * one workgroup = TM x 128 tile of one group (TM = 128: 4 waves, TM = 256: 8 waves), every wave a 64 x 64 block: 2 x 2 MFMA tiles, 128
//     accumulator registers for main and cross sums, 12 MFMAs per 16-k step on 4 + 4 fragment reads -- 341 (TM 128) / 256 (TM 256) bytes per MFMA;
//   * K is walked in STAGES of 32 k = 32 channels of one tap: TM x 128 B of packed weight fragments (lane-linear, as they are) + 128 columns
//     x 128 B of shadow rows, copied global -> LDS by LDS-DMA (global_load_lds_dwordx4: no registers, no VALU), two stage buffers; EVERY
//     wave copies its share of stage s + 1 and multiplies stage s -- round 4's conv_gk16 (experiments/) used four copying and four
//     multiplying waves and four buffers and owned its CU (128 KiB of LDS, 8 x 256 registers); here the TM = 128 form is half that footprint
//     and two workgroups share a CU, one wave of each per SIMD: while one waits for its copies the other multiplies;
//   * the B image is [column][8 slots of 16 bytes] with slot ^= (column >> 1) & 7, applied through the per-lane SOURCE address of the copy
//     (LDS-DMA writes lane-linear): the 16 lanes of a ds_read_b128 group sit in 16 different columns and hit 16 different bank slots;
//   * K is split over S workgroups per tile (tiles x S ~ the workgroup slots of the chip); the tail is a reduce-scatter through the
//     workspace: the tile goes through LDS as T[column][TM channels] in two halves of 64 columns (the half that holds this part's own slab
//     last), every part publishes the other parts' column slabs write-through (sc1), counts itself in, waits -- bounded, device flag bit 1 --
//     for its siblings (dispatched back to back, they arrive within ~1 us), adds the S contributions to ITS slab in part order (a fixed
//     order: deterministic) and runs the stream-K kernel's epilogue on it (bias, residual, output activation, f32 store, shadow store,
//     the non-finite check); the last part to leave zeroes the tile's counters.
// Per accumulator the order is that of the other split kernels (hi*hi | hi*lo, lo*hi; main + cross / 2048 per part); K is cut elsewhere than in
// the stream-K kernel, so results agree with it to f32 round-off, not bit for bit; the same call is bit-reproducible.
#include "adk_common.h"
#include <atomic>
#include <cstdlib>
#include <cstring>
#include <mutex>

#ifndef ADK_WT16_DBG
#define ADK_WT16_DBG 0      // tuning builds: 1 = per-workgroup wall-clock stamps (s_memrealtime, 100 MHz) of wave 0
#endif

namespace adk {

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8w __attribute__((ext_vector_type(8)));
typedef unsigned u32x4w __attribute__((ext_vector_type(4)));
constexpr float kWtLoScale = 2048.f, kWtLoInv = 1.f / 2048.f;
constexpr int kWtCounters = 4096;      // the arrival counters at the end of the stream-K workspace (shared with conv_gv16: zero between launches)

struct WtArgs {
    float* ws; unsigned ws_bytes;
    unsigned* counters;   // [tiles][2]: parts of the tile that have published / have left (0 between launches)
    int S;                // K parts per tile (1, 2, 4 or 8)
    int G;                // tiles * S work items
    int m_tiles, n_tiles, nstages;
    int spt;              // 32-channel stages per tap = cin_g / 32
    int kbytes_mt32;      // bytes of packed fragments per 32-row m-tile = ktot * 128
    int mt32_per_g;
    float inv_t_out;
    int* err;
};

#if ADK_WT16_DBG & 1
// 0 entry, 1 prologue copies issued, 2 first stage landed, 3 loop done, 7 slabs published (stores drained), 4 siblings arrived, 5 own slab finished, 6 exit
__device__ unsigned long long g_wt_trace[1024 * 8];
#define WT_STAMP(i) do { if (wave == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
                         if (lane == 0 && r < 1024) g_wt_trace[r * 8 + (i)] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)
#else
#define WT_STAMP(i) do { } while (0)
#endif

#define WT_DMA16(gptr, m0val) do { unsigned m0_keep_; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                                                            : "=&s"(m0_keep_) : "v"(gptr), "s"(m0val) : "memory"); } while (0)      /* M0 is the compiler's: put back */

__device__ __forceinline__ int wt_fast_div(int n, int d, float inv_d) {
    int q = (int)(__int2float_rn(n) * inv_d);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; }
    return q;
}

template <int N> __device__ __forceinline__ void wt_wait_barrier() {
    static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 18) asm volatile("s_waitcnt vmcnt(18) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else if constexpr (N == 24) asm volatile("s_waitcnt vmcnt(24) lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else static_assert(N == 0, "add the count");
}

// TM = rows of the tile (128 / 256); NW = TM / 32 waves, wave (wm, wn) = rows wm * 64 .., columns wn * 64 ..; NB = stage buffers (NB - 1 stages in flight)
template <int TM, int NB>
__global__ __launch_bounds__(TM * 2, (NB * (TM * 128 + 128 * 128) <= 80 * 1024) ? 2 : (TM == 256 ? 2 : 1)) void conv_wt16_kernel(ConvArgs a, WtArgs wt) {
    constexpr int NW = TM / 32, NT = 64 * NW;
    constexpr int A_ST = TM * 128;                          // bytes of weight fragments per stage (TM rows x 32 k x (hi + lo))
    constexpr int B_ST = 128 * 128;                         // 128 columns x 32 channels of [8 hi][8 lo] groups
    constexpr int BUF = A_ST + B_ST;                        // one stage buffer: A then B
    constexpr int BI = 128 / (8 * NW);                      // B copy instructions per wave and stage (8 columns each): 4 / 2
    constexpr int TS = TM * 4 + 16;                         // row stride of the tail's tile image T[column][TM floats]
    constexpr int IPS = 4 + BI;                             // copy instructions per wave and stage
    static_assert(64 * TS <= NB * BUF, "the tail's half tile fits the stage buffers");
    static_assert(NB >= 2 && NB <= 4, "2 .. 4 stage buffers");
    extern __shared__ __attribute__((aligned(16))) unsigned char wts[];      // [NB][BUF]; the tail's T afterwards
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-contiguous work items: the parts of a tile, and the tiles of a group, share an L2
    const int per_xcd = (wt.G + 7) >> 3;
    const int r = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (r >= wt.G) return;
    WT_STAMP(0);
    const int tile = r / wt.S, part = r - tile * wt.S;
    const int mt = tile % wt.m_tiles;
    const int rest = tile / wt.m_tiles;
    const int nt = rest % wt.n_tiles;
    const int g = rest / wt.n_tiles;
    const int s0 = (int)(((long long)part * wt.nstages) / wt.S), s1 = (int)(((long long)(part + 1) * wt.nstages) / wt.S);
    const int ns = s1 - s0;

    typedef unsigned char __attribute__((address_space(3)))* lds_u8_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_u8_t)wts;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- copy sources.  A: this wave copies 32-row m-tile `wave` of the tile: 4 KiB per stage, contiguous in the packed weights ----
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wfrag) +
                                (size_t)(g * wt.mt32_per_g + mt * NW + wave) * (size_t)wt.kbytes_mt32 + lane16;
    // B: instruction p of this wave = columns 8 * (BI * wave + p) .. + 7 of the tile, 128 bytes (32 channels: four [8 hi][8 lo] groups) each;
    // lane -> (column, 16-byte slot), slot ^= (column >> 1) & 7
    const unsigned row_bytes = (unsigned)a.in_ch * 4u;
    const unsigned ring_bytes = (unsigned)a.in_rows * row_bytes;
    const unsigned dil_bytes = (unsigned)a.dilation * row_bytes;
    const unsigned char* bsrc[BI];
    unsigned rowb[BI];
#pragma unroll
    for (int p = 0; p < BI; ++p) {
        const int col = 8 * (BI * wave + p) + (lane >> 3);
        int n = nt * 128 + col;
        if (n >= a.n_total) n = a.n_total - 1;              // (columns past the end: computed on a valid column, never stored)
        const int b = wt_fast_div(n, a.t_out, wt.inv_t_out), t = n - b * a.t_out;
        int row = a.in_row0 + t * a.stride;
        if (row >= a.in_rows) row -= a.in_rows;
        rowb[p] = (unsigned)row * row_bytes;
        const unsigned slot = (unsigned)((lane & 7) ^ ((col >> 1) & 7));
        bsrc[p] = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * ring_bytes + (size_t)(a.in_choff + g * a.in_gstride) * 4u + slot * 16u;
    }
    int is_ = 0;                                            // stage (relative to s0) whose copies are issued next; wave-uniform
    int tap_i = s0 / wt.spt, blk_i = s0 - tap_i * wt.spt;   // ... its tap and 32-channel block
    auto issue_stage = [&]() __attribute__((always_inline)) {
        const unsigned buf = lds0 + (unsigned)(is_ % NB) * BUF;
#pragma unroll
        for (int p = 0; p < 4; ++p)
            WT_DMA16(wsrc + (size_t)(s0 + is_) * 4096u + (size_t)p * 1024u, buf + (unsigned)wave * 4096u + (unsigned)p * 1024u);
        unsigned tb = (unsigned)tap_i * dil_bytes;
#pragma unroll
        for (int p = 0; p < BI; ++p) {
            unsigned rb = rowb[p] + tb;
            if (rb >= ring_bytes) rb -= ring_bytes;
            if (rb >= ring_bytes) rb -= ring_bytes;
            WT_DMA16(bsrc[p] + rb + (unsigned)blk_i * 128u, buf + A_ST + (unsigned)(BI * wave + p) * 1024u);
        }
        ++is_;
        if (++blk_i == wt.spt) { blk_i = 0; ++tap_i; }
    };

    f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }

#pragma unroll
    for (int q = 0; q < NB - 1; ++q) issue_stage();         // (a part has >= 4 stages)
    WT_STAMP(1);
    // this lane's fragment addresses (bytes from the start of a stage buffer)
    const unsigned a_off = (unsigned)(wm * 2) * 4096u + lane16;                           // + i * 4096 + (2 * st + half) * 1024
    const unsigned x8 = (unsigned)((l31 >> 1) & 7);
    const unsigned b_off = A_ST + (unsigned)(wn * 64 + l31) * 128u;                       // + j * 32 * 128 + slot * 16

    for (int sg = 0; sg < ns; ++sg) {
        // my copies of stage sg have landed (those of the <= NB - 2 stages behind it may stay in flight: loads return in order); past the
        // barrier everybody's have, and every wave is done reading stage sg - 1, whose buffer the copies of stage sg + NB - 1 go to
        const int later = ns - 1 - sg < NB - 2 ? ns - 1 - sg : NB - 2;
        if (later <= 0) wt_wait_barrier<0>();
        else if (later == 1) wt_wait_barrier<IPS>();
        else wt_wait_barrier<2 * IPS>();
        if (sg == 0) WT_STAMP(2);
        if (sg + NB - 1 < ns) issue_stage();
        const unsigned char* Sb = wts + (size_t)(sg % NB) * BUF;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f16x8w ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const f16x8w*>(Sb + a_off + i * 4096 + (2 * st) * 1024);
                al[i] = *reinterpret_cast<const f16x8w*>(Sb + a_off + i * 4096 + (2 * st + 1) * 1024);
            }
            const unsigned hs = ((unsigned)(4 * st + 2 * lh) ^ x8) * 16u, ls = ((unsigned)(4 * st + 2 * lh + 1) ^ x8) * 16u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const f16x8w*>(Sb + b_off + j * 32 * 128 + hs);
                bl[j] = *reinterpret_cast<const f16x8w*>(Sb + b_off + j * 32 * 128 + ls);
            }
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bh[j], acc[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[i], bl[j], accx[i][j], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) accx[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[i], bh[j], accx[i][j], 0, 0, 0);
        }
    }
    WT_STAMP(3);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(accx[i][j][e], kWtLoInv, acc[i][j][e]);

    // ---- tail: two halves of 64 columns through T[64 columns][TM channels] (TS-byte rows) in the stage buffers; the half that holds this
    // part's slab LAST, so that the slab can be finished straight from T once the siblings have arrived ----
    const int S = wt.S;
    const int W = 128 / S;                                  // columns of a part's slab
    const int my0 = part * W;                               // first column of mine
    const int own_half = my0 >> 6;
    const __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc(wt.ws, 0, wt.ws_bytes, 0x00020000);
    const unsigned slab_bytes = (unsigned)W * (unsigned)(TM * 4);
    constexpr int CPT = TM / 4;                             // 16-byte pieces per column
    constexpr int CPP = NT / CPT;                           // columns per pass of the publishing loop
    for (int pass = 0; pass < 2; ++pass) {
        const int half = pass == 0 ? 1 - own_half : own_half;
        __syncthreads();                                    // every wave is done with what the buffers held (the loop's operands / the previous half)
        if (wn == half) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    unsigned char* trow = wts + (size_t)(j * 32 + l31) * TS + (size_t)(wm * 64 + i * 32 + 4 * lh) * 4;
#pragma unroll
                    for (int qd = 0; qd < 4; ++qd)
                        *reinterpret_cast<float4*>(trow + qd * 32) = make_float4(acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]);
                }
        }
        __syncthreads();
        if (S > 1) {
            // my contribution to the OTHER parts' slabs that lie in this half: whole columns, 16 bytes per lane along the channel axis
            for (int q = 0; q < S; ++q) {
                if (q == part || ((q * W) >> 6) != half) continue;
                const unsigned dst = ((unsigned)((tile * S + q) * S + part)) * slab_bytes;
                const int c_lo = q * W - half * 64;          // the slab's first column inside T
                for (int c = tid / CPT; c < W; c += CPP) {
                    const u32x4w v = *reinterpret_cast<const u32x4w*>(wts + (size_t)(c_lo + c) * TS + (size_t)(tid % CPT) * 16);
                    __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_ws, dst + (unsigned)c * (unsigned)(TM * 4) + (unsigned)(tid % CPT) * 16u, 0, 16 /* sc1 */);
                }
            }
        }
    }
    if (S > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        WT_STAMP(7);
        if (tid == 0) {
            __hip_atomic_fetch_add(wt.counters + 2 * tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(wt.counters + 2 * tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)S) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 20)) { atomicOr(wt.err, 2); break; }              // never hang the device
            }
        }
        __syncthreads();
    }
    WT_STAMP(4);
    // ---- my slab (T holds its half): a thread = 8 channels (one shadow group) of one column ----
    constexpr int GPC = TM / 8;                             // 8-channel groups per column
    constexpr int CPQ = NT / GPC;                           // columns per pass
    const int cg = tid % GPC;
    const int ml = mt * TM + 8 * cg;                        // channel within the group
    const int mg = g * a.cout_g + ml;
    float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;
    if (a.bias) { bias0 = *reinterpret_cast<const float4*>(a.bias + mg); bias1 = *reinterpret_cast<const float4*>(a.bias + mg + 4); }
    int ph = 0, ocol = mg;
    if (a.up > 1) { ph = mg / a.cout_real; ocol = mg - ph * a.cout_real; }
    bool bad = false;
    const int t0c = my0 - own_half * 64;                    // my slab's first column inside T
    for (int c = tid / GPC; c < W; c += CPQ) {
        const int n = nt * 128 + my0 + c;
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        u32x4w pv[2 * 8];
        if (S > 1) {
#pragma unroll
            for (int sp = 0; sp < 8; ++sp)
                if (sp < S && sp != part) {
                    const unsigned src = ((unsigned)((tile * S + part) * S + sp)) * slab_bytes + (unsigned)c * (unsigned)(TM * 4) + (unsigned)cg * 32u;
                    pv[2 * sp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, src, 0, 16 /* sc1 */);
                    pv[2 * sp + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, src + 16u, 0, 16 /* sc1 */);
                }
        }
        const float4 own0 = *reinterpret_cast<const float4*>(wts + (size_t)(t0c + c) * TS + (size_t)cg * 32);
        const float4 own1 = *reinterpret_cast<const float4*>(wts + (size_t)(t0c + c) * TS + (size_t)cg * 32 + 16);
#pragma unroll
        for (int sp = 0; sp < 8; ++sp)
            if (sp < S) {
                if (sp == part) {
                    t0.x += own0.x; t0.y += own0.y; t0.z += own0.z; t0.w += own0.w;
                    t1.x += own1.x; t1.y += own1.y; t1.z += own1.z; t1.w += own1.w;
                } else {
                    const u32x4w u = pv[2 * sp], v = pv[2 * sp + 1];
                    t0.x += __uint_as_float(u.x); t0.y += __uint_as_float(u.y); t0.z += __uint_as_float(u.z); t0.w += __uint_as_float(u.w);
                    t1.x += __uint_as_float(v.x); t1.y += __uint_as_float(v.y); t1.z += __uint_as_float(v.z); t1.w += __uint_as_float(v.w);
                }
            }
        if (n >= a.n_total) continue;
        // the stream-K kernel's epilogue (sk_epilogue_lds), 8 channels of one column at a time: bias, residual, output activation, store, shadow
        bad |= !(fabsf(t0.x) <= 3.0e38f) | !(fabsf(t0.y) <= 3.0e38f) | !(fabsf(t0.z) <= 3.0e38f) | !(fabsf(t0.w) <= 3.0e38f) |
               !(fabsf(t1.x) <= 3.0e38f) | !(fabsf(t1.y) <= 3.0e38f) | !(fabsf(t1.z) <= 3.0e38f) | !(fabsf(t1.w) <= 3.0e38f);
        const int bb = wt_fast_div(n, a.t_out, wt.inv_t_out), t = n - bb * a.t_out;
        if (a.bias) {
            t0.x += bias0.x; t0.y += bias0.y; t0.z += bias0.z; t0.w += bias0.w;
            t1.x += bias1.x; t1.y += bias1.y; t1.z += bias1.z; t1.w += bias1.w;
        }
        if (a.res) {
            int rrow = a.res_cursor + t;
            if (rrow >= a.res_rows) rrow -= a.res_rows;
            const float* resp = a.res + ((size_t)bb * a.res_rows + rrow) * a.res_ch + a.res_choff + g * a.res_gstride + ml;
            const float4 r0 = *reinterpret_cast<const float4*>(resp), r1 = *reinterpret_cast<const float4*>(resp + 4);
            t0.x += r0.x; t0.y += r0.y; t0.z += r0.z; t0.w += r0.w;
            t1.x += r1.x; t1.y += r1.y; t1.z += r1.z; t1.w += r1.w;
        }
        float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        if (a.act_out != ADK_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = act_apply(x[e], a.act_out, 0.f);
        }
        int orow = a.out_cursor + t * a.up + ph;
        if (orow >= a.out_rows) orow -= a.out_rows;
        const size_t oidx = ((size_t)bb * a.out_rows + orow) * a.out_ch + a.out_choff + ocol;
        *reinterpret_cast<float4*>(a.out + oidx) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(a.out + oidx + 4) = make_float4(x[4], x[5], x[6], x[7]);
        if (a.out_sh) {
            f16x8w hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = act_apply(x[e], a.sh_act, a.sh_slope);
                const _Float16 h = (_Float16)y;
                hi[e] = h;
                lo[e] = (_Float16)((y - (float)h) * kWtLoScale);
            }
            unsigned char* sp_ = reinterpret_cast<unsigned char*>(a.out_sh + ((size_t)bb * a.out_rows + orow) * a.out_ch + a.out_choff) + (size_t)(ocol >> 3) * 32;
            *reinterpret_cast<f16x8w*>(sp_) = hi;
            *reinterpret_cast<f16x8w*>(sp_ + 16) = lo;
        }
    }
    if (bad) atomicOr(wt.err, 8);
    if (S > 1) {
        WT_STAMP(5);
        __syncthreads();                                    // every thread of this part has read the other parts' slabs
        if (tid == 0) {
            const unsigned gone = __hip_atomic_fetch_add(wt.counters + 2 * tile + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gone == (unsigned)(S - 1)) {                // the last part to leave: all S are past their waits and their reads
                __hip_atomic_store(wt.counters + 2 * tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(wt.counters + 2 * tile + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    WT_STAMP(6);
}

std::atomic<int> g_wt{1};            // option "wt16" / ADK_WT16: 0 = never, 1 (default) = where it is preferred, 2 = wherever it is supported (tests)
std::atomic<int> g_wt_tm{128};       // option "wt16_rows" / ADK_WT16_TM: tile rows, 128 (two workgroups per CU) or 256 (one)
std::atomic<int> g_wt_nb{2};         // option "wt16_buffers" / ADK_WT16_NB: stage buffers (2 .. 4; bytes in flight per workgroup = (NB - 1) stages)
std::atomic<int> g_wt_min_work{2048};   // ADK_WT16_MIN_WORK: tiles(128 x 128) x stages from which the kernel is preferred
std::once_flag g_wt_once;
void wt_read_env() {
    std::call_once(g_wt_once, [] {
        const char* e = getenv("ADK_WT16"); if (e) g_wt.store(atoi(e));
        e = getenv("ADK_WT16_TM"); if (e && (atoi(e) == 128 || atoi(e) == 256)) g_wt_tm.store(atoi(e));
        e = getenv("ADK_WT16_NB"); if (e && atoi(e) >= 2 && atoi(e) <= 4) g_wt_nb.store(atoi(e));
        e = getenv("ADK_WT16_MIN_WORK"); if (e && atoi(e) >= 0) g_wt_min_work.store(atoi(e));
    });
}

}  // namespace

#if ADK_WT16_DBG & 1
extern "C" int adk_debug_wt_trace(unsigned long long* out, int n) {
    if (n > 1024 * 8) n = 1024 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wt_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

int conv_wt16_set_option(const char* name, int value) {
    wt_read_env();
    if (!strcmp(name, "wt16")) { g_wt.store(value < 0 ? 0 : value); return 0; }
    if (!strcmp(name, "wt16_rows")) { if (value != 128 && value != 256) return -1; g_wt_tm.store(value); return 0; }
    if (!strcmp(name, "wt16_buffers")) { if (value < 2 || value > 4) return -1; g_wt_nb.store(value); return 0; }
    if (!strcmp(name, "wt16_min_work")) { g_wt_min_work.store(value < 0 ? 0 : value); return 0; }
    return 1;
}

static int wt_rows(const ConvArgs& a) {
    wt_read_env();
    const int tm = g_wt_tm.load();
    return (tm == 256 && a.cout_g % 256 == 0) ? 256 : 128;
}

bool conv_wt16_supported(const ConvArgs& a) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!a.in_sh || !a.wfrag || !al16(a.in_sh) || !al16(a.wfrag) || !al16(a.out)) return false;
    if (a.cin_g % 32 || a.cout_g % 128 || a.cout_real % 8 || a.n_total < 1 || a.n_total >= (1 << 24)) return false;
    if (a.ktot % 64) return false;                                                                    // (whole 64-k groups in the packed weights: no zero tail to skip)
    if (a.in_ch % 8 || a.in_choff % 8 || a.in_gstride % 8 || a.out_ch % 8 || a.out_choff % 8) return false;       // whole 8-channel shadow groups
    if (a.up > 1 && a.cout_real % 8) return false;
    if ((unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull >= 0x80000000ull) return false;
    if (a.bias && !al16(a.bias)) return false;
    if (a.res && (a.res_ch % 4 || a.res_choff % 4 || a.res_gstride % 4 || !al16(a.res))) return false;
    const long long tiles = (long long)(a.cout_g / 128) * ((a.n_total + 127) / 128) * a.groups;
    return 2 * tiles <= kWtCounters;          // two counters per tile (arrived / left)
}

bool conv_wt16_preferred(const ConvArgs& a) {
    wt_read_env();
    const int mode = g_wt.load();
    if (!mode || !conv_wt16_supported(a)) return false;
    if (mode >= 2) return true;
    // enough work for ~480 workgroups of >= 8 stages: the wide grouped convs of a v1 vocoder's first stage at >= 128 streams (60 tiles x
    // 88 stages at 256); the smaller stream-K launches (a few tiles, K <= 1792) stay where they are
    const long long tiles = (long long)(a.cout_g / 128) * ((a.n_total + 127) / 128) * a.groups;
    return tiles * (a.ktot / 32) >= g_wt_min_work.load();
}

const char* conv_wt16_name(const ConvArgs& a) { return wt_rows(a) == 256 ? "conv_wt16<256x128>" : "conv_wt16<128x128>"; }

int launch_conv_wt16(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    if (a.n_total == 0) return ADK_OK;
    if (!conv_wt16_supported(a)) return fail(ADK_ERR_STATE, "conv_wt16: unsupported arguments");
    const int TM = wt_rows(a);
    WtArgs wt;
    wt.m_tiles = a.cout_g / TM;
    wt.n_tiles = (a.n_total + 127) / 128;
    wt.nstages = a.ktot / 32;
    wt.spt = a.cin_g / 32;
    wt.kbytes_mt32 = a.ktot * 128;
    wt.mt32_per_g = a.cout_g / 32;
    wt.inv_t_out = 1.0f / (float)a.t_out;
    const int tiles = wt.m_tiles * wt.n_tiles * a.groups;
    // K parts per tile: a power of two <= 8, tiles x S within the workgroup slots of the chip (two per CU for 128 rows, one for 256), >= 4 stages per part
    const int slots = (TM == 128 && g_wt_nb.load() == 2) ? 512 : 256;      // two workgroups per CU only with 64 KiB of LDS each
    int S = 1;
    while (S < 8 && tiles * S * 2 <= slots && wt.nstages / (S * 2) >= 4) S *= 2;
    wt.S = S; wt.G = tiles * S;
    size_t flags_offset = 0;
    const size_t need = conv_mfma_workspace_bytes(&flags_offset);
    if (!ws.ptr || ws.bytes < need || (size_t)wt.G * (size_t)(TM * 128 * 4) > flags_offset) return fail(ADK_ERR_STATE, "conv_wt16: workspace missing or too small");
    wt.ws = ws.ptr; wt.ws_bytes = (unsigned)std::min<size_t>(flags_offset, 0x7fffffffu);
    wt.counters = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws.ptr) + need - kWtCounters * sizeof(unsigned));
    wt.err = conv_err_word(a);
    ConvArgs b = a;
    b.in = a.in_sh;
    const unsigned grid = (unsigned)((wt.G + 7) / 8 * 8);
    int nb = g_wt_nb.load();
    if (TM == 256 && nb > 3) nb = 3;                       // 4 x 48 KiB does not fit
    const size_t lds = (size_t)nb * (TM * 128 + 128 * 128);
    auto go = [&](auto kern, int slot) -> int {
        static bool attr_dev[8][kMaxDevices] = {};
        bool& attr = attr_dev[slot][current_device()];
        if (!attr) {
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr = true;
        }
        hipLaunchKernelGGL(kern, dim3(grid), dim3(TM * 2), lds, s, b, wt);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    if (TM == 256) return nb == 2 ? go(conv_wt16_kernel<256, 2>, 0) : go(conv_wt16_kernel<256, 3>, 1);
    if (nb == 2) return go(conv_wt16_kernel<128, 2>, 2);
    if (nb == 3) return go(conv_wt16_kernel<128, 3>, 3);
    return go(conv_wt16_kernel<128, 4>, 4);
}

}  // namespace adk
