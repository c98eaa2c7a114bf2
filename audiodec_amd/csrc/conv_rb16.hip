// conv_rb16 -- a whole RESIDUAL CHAIN in one launch, activations resident in LDS (split-f16 operands on the f16 matrix cores).
//
// The reference runs a residual block as a Python loop over modules, every intermediate tensor going through memory:
//   HiFiGANResidualBlock.inference  models/vocoder/modules/residual_block.py:99-105
//       for idx: xt = convs1[idx](act(x)); xt = convs2[idx](act(xt)); x = xt + x
//   EncoderBlock / DecoderBlock     models/autoencoder/modules/{encoder.py:76-81, decoder.py:73-78} over
//   CausalResidualUnit.inference    models/autoencoder/modules/residual_unit.py:78-81      x + conv2(act(conv1(act(x))))
// i.e. a chain of `units` pairs (conv A: K taps, dilation d_u; conv B: K taps or 1x1, dilation 1; + residual), each conv
// carrying its own causal history (layers/conv_layer.py:153-156).  The per-op lowering of this repository kept that shape:
// 2 * units launches per block (units launches with conv_rl16<FUSE>), each of which stages its input rows from HBM / L2,
// multiplies, and stores -- 13 us of matrix-core work in a 32 us span per launch (profiles/r2_sk16_timeline.md 5-6).
//
// Here ONE workgroup owns `spw` streams of one group for the whole chain:
//   * the chain input (new rows + history of conv 0) is staged ONCE into LDS, activated and split into f16 hi / lo halves
//     (the B-operand layout of conv_rl16.hip: row = [C halfs hi][C halfs lo][16 B pad]);
//   * every conv is the rows-in-LDS implicit GEMM of conv_rl16 (weights stream from L2 in MFMA-fragment order, two chunks
//     ahead; B fragments are ds_read_b128 at a shifted row); its result stays in REGISTERS, gets bias / residual there,
//     and after a barrier is written back over the same LDS rows, activated and split, as the next conv's input;
//   * the residual x of a unit is read back from its ring at the END of the unit's second conv, by the very lane that stored it
//     two convs earlier (same column, same channels: a lane always sees its own stores) -- 32 registers per wave would
//     otherwise be pinned through both MFMA loops, and the kernel has none to spare at 3 waves per SIMD;
//   * the causal history of conv k+1 (rows its input ring received in EARLIER calls) is fetched from the ring while the
//     epilogue of conv k runs, and written in front of the new rows;
//   * the intermediate h of a unit goes to HBM only as far as later calls need it as history: its last `keep` rows
//     (keep = the ring's history length: 10 of 100..300 rows for the K11 blocks, none for a 1x1 second conv); unit outputs
//     are stored in full (the next unit's residual, and the last `keep` rows are later calls' history).
// Columns of the implicit GEMM are (stream, step) pairs packed densely (n = s * T + t), so several streams of a short
// frame share a 32-column MFMA tile (128-channel layers: 2 streams x 25 steps per workgroup).
// Per output element the operations and their order are exactly those of conv_rl16 (k ascending, acc0 + acc1/2048, + bias,
// + residual): the result is BIT-IDENTICAL to the per-op path (tests/test_gpu_b256.py::test_residual_chains_are_bit_identical).
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

#ifndef ADK_RB16_PF32
#define ADK_RB16_PF32 2            // weight prefetch distance in 16-k steps of the variants in use (tuning builds: -DADK_RB16_PF32=... etc.)
#endif
#ifndef ADK_RB16_PF64
#define ADK_RB16_PF64 2
#endif
#ifndef ADK_RB16_PF128
#define ADK_RB16_PF128 3           // (with the loads pinned: 2 / 3 / 4 / 5 / 6 -> 259.3 / 262.0 / 259.7 / 257.7 / ~250 k frames/s on one box; 6 was the unpinned choice)
#endif
#ifndef ADK_RB16_PIN_LOADS
#define ADK_RB16_PIN_LOADS 1
#endif
#ifndef ADK_RB16_ASYNC_TOUCH
#define ADK_RB16_ASYNC_TOUCH 1     // 0: the L2 warm-up touches as volatile loads (each one waited for), as measured in profiles/r3_rb16_timeline.md sections 1-3
#endif
#ifndef ADK_RB16_RING_SLOTS32
#define ADK_RB16_RING_SLOTS32 4    // LDS weight ring of the 32-channel chains: slots of 4 KiB = one group of two 16-k steps (round 4; see rb_mfma_ring)
#endif
#ifndef ADK_RB16_HIST_LATE
#define ADK_RB16_HIST_LATE 1
#endif
#ifndef ADK_RB16_HIST_LATE128
#define ADK_RB16_HIST_LATE128 0     // 128-channel variants: the next conv's history rows (up to 64 registers of pieces per thread) requested behind the finish pass
#endif
#ifndef ADK_RB16_EARLY64
#define ADK_RB16_EARLY64 true      // 64-channel ring variant: residual fetched under the second conv's MFMAs (true) or behind them
#endif
#ifndef ADK_RB16_DBG
#define ADK_RB16_DBG 0      // tuning builds only: 1 = per-workgroup wall-clock stamps (s_memrealtime, 100 MHz) at every phase boundary of wave 0;
                            // knock-outs (results are garbage): 2 = no weight loads inside the MFMA loops, 4 = no MFMAs, 8 = one B-fragment read per loop,
                            // 16 = every workgroup reads the weight stream from a different offset (are synchronised readers of the same lines the problem?),
                            // 32 = a barrier every 4 steps of the MFMA loops (waves of a workgroup in lockstep: their identical loads merge in L1)
#endif

namespace adk {

#if ADK_RB16_DBG & 1
// [launch slot 0..15][workgroup 0..1023][stamp 0..31]: stamp 0 = entry, 1 = input staged, then per conv k: 2+4k = MFMA loop done,
// 3+4k = results in registers / ring stores issued, 4+4k = passed the first barrier, 5+4k = LDS written + second barrier passed
__device__ unsigned long long g_rb_trace[16 * 1024 * 32];
extern "C" int adk_debug_rb_trace(unsigned long long* out, int n) {
    if (n > 16 * 1024 * 32) n = 16 * 1024 * 32;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_rb_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
static int g_rb_launch = 0;
#define RB_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (wave == 0 && blockIdx.x < 1024) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (lane == 0) g_rb_trace[((size_t)(r.dbg_slot & 15) * 1024 + blockIdx.x) * 32 + (i)] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define RB_STAMP(i) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4s __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int kRbMaxConvs = 8;
constexpr int kRbMaxHist = 56;           // rows of causal history a conv of the chain may need ((K-1) * dilation: 54 for K7 d9, 50 for K11 d5)
constexpr int kRbTouchSink = 256;      // bytes at the end of the dynamic LDS that the L2 warm-up touches (LDS-DMA loads nobody reads) land in
constexpr float kRbLoScale = 2048.f, kRbLoInv = 1.f / 2048.f;

struct RbNode {                          // a tensor of the chain: where its rows live in HBM (a state ring)
    float* base; int rows, ch, cursor, choff, gstride;      // row of step t of stream b: base + (b * rows + (cursor + t) mod rows) * ch + choff + g * gstride
    int keep;                            // rows of history later calls read from this ring (intermediate nodes only)
};
struct RbConv { const float* wfrag; const float* bias; int dil, hist, ksteps; unsigned w_bytes; };
struct RbArgs {
    RbNode node[kRbMaxConvs + 1];        // node k = input of conv k, node k+1 = its output; node 2u = residual of conv 2u+1
    RbConv conv[kRbMaxConvs];
    int n_convs, batch, t, groups;
    int spw;                             // streams per workgroup
    int hm;                              // history rows in front of a stream's new rows in LDS (max over convs)
    int rps;                             // LDS rows per stream = hm + t
    int n_tiles;                         // 32-column tiles of a full workgroup (ceil(spw * t / 32))
    float slope; int* err;
    int warm;                            // touch the next conv's weights / the history rows ahead of use (launches of one workgroup per CU)
    int n_main;                          // workgroups that compute: blocks [0, n_main)
    int helpers;                         // > 0: blocks [8 * ceil(n_main / 8), ... + 8 * helpers) only warm the L2s (few streams; see the kernel head)
    int helper_convs;                    // ... with the weights of convs [0, helper_convs)
    int dbg_slot;
};

template <int ACT>
__device__ __forceinline__ float rb_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

__device__ __forceinline__ f16x8 rb_as_f16x8(const u32x4s& v) {
    union { u32x4s u; f16x8 h; } c; c.u = v; return c.h;
}

// 8 consecutive channels of one row: activation, split into hi / lo halves, into the row's LDS slots
template <int C, int ACT>
__device__ __forceinline__ void rb_put8(unsigned char* dst, const float4& u, const float4& v, float slope) {
    const float x[8] = {u.x, u.y, u.z, u.w, v.x, v.y, v.z, v.w};
    f16x8 hi, lo;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        const float y = rb_act<ACT>(x[j], slope);
        const _Float16 h = (_Float16)y;
        hi[j] = h;
        lo[j] = (_Float16)((y - (float)h) * kRbLoScale);
    }
    *reinterpret_cast<f16x8*>(dst) = hi;
    *reinterpret_cast<f16x8*>(dst + 2 * C) = lo;
}

// One conv of the chain for this wave's work item (m-tile, NA n-tiles): the MFMA sequence of conv_rl16_kernel per n-tile,
// straight-line (an n-tile past the last column is computed on clamped addresses and never stored).
// x[j]: this lane's B-fragment address for tap 0, chunk 0 of n-tile j; dil_rs = dilation * row stride.
template <int C, int TAPS, int PF, int LS, int NA, int NTW>
__device__ __forceinline__ void rb_mfma(const unsigned char* const (&x)[NTW], int dil_rs,
                                        const __amdgpu_buffer_rsrc_t rsrc_w, unsigned lane16, unsigned wbase,
                                        u32x4s (&ah)[PF + 1], u32x4s (&al)[PF + 1], f32x16 (&m)[NTW], f32x16 (&c)[NTW]) {
    constexpr int CH = C / 16, STEPS = TAPS * CH;
#pragma unroll
    for (int s = 0; s < STEPS; ++s) {
        if (s + PF < STEPS && !(ADK_RB16_DBG & 2)) {
            ah[(s + PF) % (PF + 1)] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, wbase + (unsigned)(s + PF) * 2048u, 0);
            al[(s + PF) % (PF + 1)] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16 + 1024u, wbase + (unsigned)(s + PF) * 2048u, 0);
#if ADK_RB16_PIN_LOADS
            // keep the loads HERE: left alone the scheduler sinks each of them to two or three MFMAs in front of its first use (shorter
            // live ranges), i.e. the prefetch distance collapses from PF steps to ~100 cycles and every step waits out an L2 round trip
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
        const int tap = s / CH, ch = s - tap * CH;
        const int off = (ADK_RB16_DBG & 8) ? 0 : tap * dil_rs + 32 * ch;
        const f16x8 Ah = rb_as_f16x8(ah[(ADK_RB16_DBG & 2) ? 0 : s % (PF + 1)]), Al = rb_as_f16x8(al[(ADK_RB16_DBG & 2) ? 0 : s % (PF + 1)]);
        f16x8 bh[NA], bl[NA];
#pragma unroll
        for (int j = 0; j < NA; ++j) {
            bh[j] = *reinterpret_cast<const f16x8*>(x[j] + off);
            bl[j] = *reinterpret_cast<const f16x8*>(x[j] + off + 2 * C);
        }
        // per accumulator the order is hi*hi | hi*lo, lo*hi -- the same as conv_rl16; the n-tiles are interleaved so that no MFMA
        // waits for the one before it
#pragma unroll
        for (int j = 0; j < NA; ++j) m[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh[j], m[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NA; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl[j], c[j], 0, 0, 0);
#pragma unroll
        for (int j = 0; j < NA; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh[j], c[j], 0, 0, 0);
        // LS > 0: the waves of the workgroup re-align every LS steps (a bare s_barrier: no memory is handed over).  Measured on the
        // 128-channel chains -- 4 waves streaming four different 180 KiB weight blocks -- the MFMA loop takes HALF the time when the
        // waves (and with them all workgroups of the launch) walk the weights in step: 38 -> 19 us per conv (profiles/r3_rb16_timeline.md)
        if (LS > 0 && (s % LS) == LS - 1 && s + 1 < STEPS) __builtin_amdgcn_s_barrier();
    }
}

// ---- The same loop with the weights coming through an LDS RING that the whole workgroup shares (round 4) ----
// rb_mfma above has every wave pull the fragments of its m-tile from L2 into registers: at 32 channels the four waves of a workgroup
// load the SAME 2 KiB per step, at 64 channels two waves each -- 45 / 90 KB per conv and wave for two or three n-tiles of work, ~13.5 TB/s
// of L2 reads chip-wide, which is what keeps these loops at ~60 % of their matrix-core bound (profiles/r3_rb16_timeline.md, knock-outs).
// Here a fragment crosses the L2 -> CU path ONCE per workgroup: the weights of a conv are walked in GROUPS of 4 KiB (32 channels: two 16-k
// steps of the one m-tile; 64 channels: one step of both m-tiles), every wave copies one 1 KiB piece of a group global -> LDS by LDS-DMA
// (no registers, no VALU), WR slots of 4 KiB form a ring, and all waves read their A fragments from the slot (ds_read_b128, lane-linear:
// conflict-free).  Hand-over per group, counted by hand because the compiler does not see the DMA:
//     s_waitcnt vmcnt(n)   this wave's piece of group i has landed (n = its pieces of the groups behind i still in flight;
//                          loads return in order, and whatever else the compiler has in flight only makes the wait stricter)
//     s_barrier            ... and so has every other wave's; and every wave is done READING group i - 1
//     DMA group i + WR - 1 into the slot group i - 1 just left (all waves are past their reads of it: they consumed them in MFMAs
//                          issued before they arrived at this barrier)
//     ds_read A (slot i), ds_read B (rows), MFMAs of the group's steps -- the k order and the accumulator order of rb_mfma: bit-identical
// Groups are numbered through the whole launch (slot = group number mod WR, `sl` carries it from conv to conv), so the first WR - 1
// groups of the NEXT conv can be requested as soon as a wave leaves the loop: they land under the epilogue.
struct RbRing {
    const unsigned char* base;      // slot 0 (generic pointer, for the ds_reads)
    unsigned lds_addr;              // ... its LDS byte address (for M0)
    int sl;                         // slot of the next group to be consumed (0 .. WR-1)
};

#define RB_DMA16(gptr, m0val) do { unsigned m0_keep_; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                                                            : "=&s"(m0_keep_) : "v"(gptr), "s"(m0val) : "memory"); } while (0)      /* M0 is the compiler's: put back */

// groups [0, min(WR - 1, NG)) of a conv: requested ahead of its loop (gsrc: this lane's source of ITS piece of group 0)
template <int C, int TAPS, int WR>
__device__ __forceinline__ void rb_ring_preload(const RbRing& rg, const unsigned char* gsrc, int wave) {
    constexpr int MT = C / 32, GS = 2 / MT, NG = TAPS * (C / 16) / GS, PD = WR - 1;
    int sl = rg.sl;
#pragma unroll
    for (int j = 0; j < PD; ++j) {
        if (j < NG) RB_DMA16(gsrc + (size_t)j * GS * 2048u, rg.lds_addr + (unsigned)sl * 4096u + (unsigned)wave * 1024u);
        sl = sl + 1 == WR ? 0 : sl + 1;
    }
}

template <int C, int TAPS, int NA, int NTW, int WR>
__device__ __forceinline__ void rb_mfma_ring(const unsigned char* const (&x)[NTW], int dil_rs, RbRing& rg, const unsigned char* gsrc, int wave, int mt,
                                             unsigned lane16, f32x16 (&m)[NTW], f32x16 (&c)[NTW]) {
    constexpr int CH = C / 16, STEPS = TAPS * CH, MT = C / 32, GS = 2 / MT, NG = STEPS / GS, PD = WR - 1;
    static_assert(MT == 1 || MT == 2, "the weight ring is built for 32 / 64 channels per group");
    static_assert(STEPS % GS == 0 && WR >= 2 && WR <= 5, "whole groups; 2..5 ring slots");
#pragma unroll
    for (int i = 0; i < NG; ++i) {
        const int behind = (NG - 1 - i) < (PD - 1) ? (NG - 1 - i) : (PD - 1);       // this wave's pieces of later groups that may stay in flight
        // (lgkmcnt(0): this wave's LDS reads of group i - 1 have RETURNED, not merely been issued, when it arrives -- the compiler
        // sinks MFMAs of the previous group below this statement, and with them the wait for their operands; a slot must not be
        // handed to the DMA while a read of it is still queued)
        if (behind <= 0) asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (behind == 1) asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else if (behind == 2) asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (i + PD < NG) {
            const int dst = rg.sl + PD >= WR ? rg.sl + PD - WR : rg.sl + PD;
            RB_DMA16(gsrc + (size_t)(i + PD) * GS * 2048u, rg.lds_addr + (unsigned)dst * 4096u + (unsigned)wave * 1024u);
        }
        __builtin_amdgcn_sched_barrier(0);
        const unsigned char* slot = rg.base + rg.sl * 4096 + lane16;
#pragma unroll
        for (int sg = 0; sg < GS; ++sg) {
            const int s = i * GS + sg;
            const int tap = s / CH, ch = s - tap * CH;
            const int off = tap * dil_rs + 32 * ch;
            const f16x8 Ah = *reinterpret_cast<const f16x8*>(slot + (sg * MT + mt) * 2048);
            const f16x8 Al = *reinterpret_cast<const f16x8*>(slot + (sg * MT + mt) * 2048 + 1024);
            f16x8 bh[NA], bl[NA];
#pragma unroll
            for (int j = 0; j < NA; ++j) {
                bh[j] = *reinterpret_cast<const f16x8*>(x[j] + off);
                bl[j] = *reinterpret_cast<const f16x8*>(x[j] + off + 2 * C);
            }
#pragma unroll
            for (int j = 0; j < NA; ++j) m[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh[j], m[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NA; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl[j], c[j], 0, 0, 0);
#pragma unroll
            for (int j = 0; j < NA; ++j) c[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh[j], c[j], 0, 0, 0);
        }
        rg.sl = rg.sl + 1 == WR ? 0 : rg.sl + 1;
    }
}

// 4 waves per workgroup: wave w works on m-tile w / WM (WM = 4 / m-tiles waves share an m-tile) and on up to NTW consecutive 32-column
// tiles of it.  SMAX = most streams a workgroup takes (sizes the register staging of the history rows).
// WPS = waves per SIMD the register budget is cut for (2: 256 registers, 3: 168); PF = weight prefetch distance in 16-k steps;
// EARLY_RES: the residual is fetched before the second conv's MFMA loop (16 registers per n-tile live through it);
// LS: lockstep interval (rb_mfma); WR: slots of the shared LDS weight ring (rb_mfma_ring), 0 = every wave streams its weights into registers.
template <int C, int ACT, int TA, int TB, int NTW, int SMAX, int WPS, int PF, bool EARLY_RES, int LS, int WR>
__global__ __launch_bounds__(256, WPS) void conv_rb16_kernel(RbArgs r) {
    constexpr int RS = 4 * C + 16;                     // LDS row stride in bytes: [C halfs hi][C halfs lo][16 B pad]
    constexpr int C8 = C / 8;
    constexpr int MT = C / 32;
    constexpr int NW = 4, NT = 64 * NW;
    constexpr int WM = NW / MT;                         // waves that share an m-tile (and its weight stream)
    constexpr bool BIAS_LDS = C < 128 && !(C == 64 && (SMAX == 2 || WR > 0));    // bias of every conv staged in LDS -- unless the LDS is needed to the last KB for a second / third workgroup per CU
    constexpr int RING_BYTES = WR * 4096;
    constexpr bool BIAS_LATE = WR > 0 && !BIAS_LDS;
    constexpr bool HIST_LATE = (((WR > 0 || ADK_RB16_HIST_LATE > 1) && WPS == 3) || (C == 128 && ADK_RB16_HIST_LATE128)) && ADK_RB16_HIST_LATE;   // 168-register ring variants: the next conv's history rows are requested BEHIND the
                                                     // finish / ring-store pass instead of in front of it (16 registers less at the epilogue's peak)   // bias fetched BEHIND the MFMA loop (16 registers per lane would otherwise live through it)
    constexpr int NHP = (SMAX * kRbMaxHist * C8 + NT - 1) / NT;      // 8-channel history pieces per thread
    extern __shared__ __attribute__((aligned(16))) unsigned char xs[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const unsigned lane16 = (unsigned)lane * 16u;
    // ---- L2 warm-up by HELPER workgroups (round 5, launches of a few workgroups).  A chain of few streams is one workgroup per (stream, group) that
    // pulls its convs' weights through ONE CU -- and between two steps they have left the L2s (a frame touches 93 MB of weights), so they come
    // from the Infinity Cache at what one CU's request path gets out of it: 720 KB per conv of the vocoder's second stage in 9.6 us = 75 GB/s
    // (profiles/r5_rb16_trace_single_stream.log), half of what the same path delivers from the L2.  The rest of the chip idles meanwhile.  So the launch
    // carries `helpers` extra workgroups per XCD that do nothing but TOUCH -- one dword per 128-byte line, into a sink nobody reads -- the weights
    // the computing workgroups of THEIR XCD will stream, conv after conv, and exit: the lines are then in that XCD's L2.  Which XCD a block runs
    // on is an observation, not a contract (block b -> XCD b % 8, MI355X_MICROARCH.md): a wrong guess costs the speed-up, nothing else -- nothing
    // that is computed depends on a helper. ----
    if ((int)blockIdx.x >= r.n_main) {
        const int pad = (r.n_main + 7) & ~7;
        if (r.helpers <= 0 || (int)blockIdx.x < pad) return;
        const int h = (int)blockIdx.x - pad, xcd = h & 7, sub = h >> 3;
        typedef unsigned char __attribute__((address_space(3)))* lds_u8h_t;
        const unsigned sink = (unsigned)(size_t)(lds_u8h_t)xs;
        unsigned seen = 0;                                   // groups whose workgroups sit on this XCD (groups <= 32 here: the host checked)
        for (int i = xcd; i < r.n_main; i += 8) seen |= 1u << (i % r.groups);
        auto touch = [&](const unsigned char* ptr) __attribute__((always_inline)) {
            unsigned m0_keep_;
            asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                         : "=&s"(m0_keep_) : "v"(ptr), "s"(sink) : "memory");
        };
        for (int k = 0; k < r.helper_convs; ++k) {
            if (k == 1) {
                // behind the first conv's weights: the HISTORY rows the later convs fetch from their state rings (rows earlier steps wrote: [cursor - hist,
                // cursor) of node k, this XCD's streams) -- every one of them is a cold round trip in the computing workgroup's epilogue otherwise
                for (int i = xcd; i < r.n_main; i += 8) {
                    const int gi = i % r.groups, s0 = (i / r.groups) * r.spw, sc = min(r.spw, r.batch - s0);
                    for (int kk = 1; kk < r.n_convs; ++kk) {
                        const RbNode& nd = r.node[kk];
                        const int hk = r.conv[kk].hist;
                        constexpr int LPR = (C * 4 + 127) / 128;
                        const int total = sc * hk * LPR;
                        for (int q = sub * NT + tid; q < total; q += r.helpers * NT) {
                            const int st_ = q / (hk * LPR), rem_ = q - st_ * hk * LPR;
                            const int rr = rem_ / LPR, li = rem_ - rr * LPR;
                            int row = nd.cursor - hk + rr;
                            if (row < 0) row += nd.rows;
                            touch(reinterpret_cast<const unsigned char*>(nd.base + ((size_t)(s0 + st_) * nd.rows + row) * nd.ch + nd.choff + gi * nd.gstride + 32 * li));
                        }
                    }
                }
            }
            const RbConv& cv = r.conv[k];
            const int nlines = MT * cv.ksteps * 16;          // 128-byte lines of one group's fragments of this conv
            for (int gg = 0; gg < r.groups; ++gg) {
                if (!((seen >> gg) & 1u)) continue;
                const unsigned char* wb = reinterpret_cast<const unsigned char*>(cv.wfrag) + (size_t)(gg * MT * cv.ksteps) * 2048u;
                for (int line = sub * NT + tid; line < nlines; line += r.helpers * NT) touch(wb + (size_t)line * 128u);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // the sink is this workgroup's LDS: nothing may still be landing when it is released
        return;
    }
    RB_STAMP(0);

    const int g = blockIdx.x % r.groups;
    const int b0 = (blockIdx.x / r.groups) * r.spw;
    const int scur = min(r.spw, r.batch - b0);
    const int T = r.t;
    const int ncols = scur * T;
    float* bias_lds = reinterpret_cast<float*>(xs + (size_t)r.spw * r.rps * RS);      // [n_convs][C]

    // ---- this wave's work item: m-tile mt, n-tiles [first, first + ntw).  The tiles of an m-tile are dealt to its WM waves as
    // evenly as they go; who gets the odd ones rotates with the workgroup, because the hardware puts wave w of EVERY workgroup
    // on the same SIMD (0, 2, 1, 3) -- without the rotation the same SIMD would carry the longer item in every co-resident
    // workgroup.  Everything below is straight-line for every wave: n-tiles the wave does not have and columns past the end are
    // computed on clamped addresses and masked where something is STORED (valid[]). ----
    const int mt = wave / WM, wi = wave - mt * WM;
    const int base = r.n_tiles / WM, rem = r.n_tiles - base * WM;
    const int pos = (wi + WM - (int)((blockIdx.x + blockIdx.x / 256u) % WM)) % WM;
    const int ntw = base + (pos < rem ? 1 : 0);
    const int first = pos * base + min(pos, rem);
    bool valid[NTW]; int sj[NTW], tj[NTW], lrow[NTW];
#pragma unroll
    for (int j = 0; j < NTW; ++j) {
        const int n = (first + j) * 32 + l31;
        valid[j] = j < ntw && n < ncols;
        const int nc = min(n, ncols - 1);
        sj[j] = nc / T; tj[j] = nc - sj[j] * T;
        lrow[j] = sj[j] * r.rps + r.hm + tj[j];
    }

    // ---- first weight fragments of conv 0: issued before anything else (their L2 round trip overlaps the staging) ----
    u32x4s ah[PF + 1], al[PF + 1];
    typedef unsigned char __attribute__((address_space(3)))* lds_u8_t;
    // the weight ring (WR > 0) sits behind the rows and the bias block; this wave's DMA piece of every group: half (hi / lo) = wave & 1,
    // (step in group, m-tile) = wave >> 1 -- 32 channels: step; 64 channels: m-tile
    const unsigned ring_off = (unsigned)r.spw * (unsigned)r.rps * (unsigned)RS + (BIAS_LDS ? (unsigned)r.n_convs * C * 4u : 0u);
    RbRing rg;
    rg.base = xs + ring_off; rg.lds_addr = (unsigned)(size_t)(lds_u8_t)xs + ring_off; rg.sl = 0;
    auto ring_src = [&](const RbConv& cv) __attribute__((always_inline)) -> const unsigned char* {
        const int q = wave >> 1;
        const int mt_ = MT == 1 ? 0 : q, sg_ = MT == 1 ? q : 0;
        return reinterpret_cast<const unsigned char*>(cv.wfrag) + ((size_t)((g * MT + mt_) * cv.ksteps + sg_)) * 2048u + (size_t)(wave & 1) * 1024u + lane16;
    };
    auto preload = [&](auto taps_c, const RbConv& cv) __attribute__((always_inline)) {
        constexpr int TAPS = decltype(taps_c)::value;
        if constexpr (WR > 0) {
            rb_ring_preload<C, TAPS, WR>(rg, ring_src(cv), wave);
        } else {
            constexpr int steps = TAPS * (C / 16);
            const __amdgpu_buffer_rsrc_t rs_ = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cv.wfrag), 0, cv.w_bytes, 0x00020000);
            const unsigned wb = (unsigned)((g * MT + mt) * cv.ksteps) * 2048u;
#pragma unroll
            for (int s = 0; s < PF; ++s) {
                if (s < steps) {
                    ah[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_, lane16, wb + (unsigned)s * 2048u, 0);
                    al[s] = __builtin_amdgcn_raw_buffer_load_b128(rs_, lane16 + 1024u, wb + (unsigned)s * 2048u, 0);
                }
            }
        }
    };
    if constexpr (WR == 0) preload(std::integral_constant<int, TA>(), r.conv[0]);

    // ---- L2 warm-up.  Every workgroup of the launch walks the same weights at about the same time, and between two calls
    // (a whole pipeline step, ~100 MB of other traffic) they have left the 4 MiB L2 of the XCD: each 2 KiB fragment pair would be
    // a first touch -- a round trip to the Infinity Cache / HBM -- with a prefetch distance of two 16-k steps.  So the lines of
    // a conv's weight block are TOUCHED one conv ahead (one dword per 128-byte line, result unused: the hardware has no
    // prefetch instruction), by the waves that will stream it; the fragment loads then find them in L2.  The same for the
    // history rows the later convs fetch from their state rings. ----
    // A touch is an LDS-DMA load issued through inline asm: the compiler does not know it is a load, so it inserts no s_waitcnt for
    // it (a volatile load was waited for in EVERY iteration of these loops, one round trip each -- ISA), and its data goes to a 256-byte
    // sink in LDS that nothing reads, not to a register the allocator might hand to somebody else while the load is in flight.
    // Nothing ever waits for a touch by name; the compiler's own s_waitcnt vmcnt(n) assume fewer loads in flight than there are
    // (loads return in order: they wait for more, never for less), and the last touches of a launch go out before the last conv's
    // MFMA loop, whose weight waits retire them long before the workgroup's LDS is released.
    const unsigned touch_m0 = rg.lds_addr + (unsigned)RING_BYTES;                      // the last kRbTouchSink bytes of the dynamic LDS
#if ADK_RB16_ASYNC_TOUCH
#define RB_TOUCH(ptr) do { unsigned m0_keep_; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0" \
                                                        : "=&s"(m0_keep_) : "v"(ptr), "s"(touch_m0) : "memory"); } while (0)      /* M0 is the compiler's: put back */
#else
#define RB_TOUCH(ptr) (void)*reinterpret_cast<const volatile unsigned*>(ptr)
#endif
    auto warm_weights = [&](const RbConv& cv) __attribute__((always_inline)) {
        // scalar loop counter, scalar block pointer, lane id recomputed: inside the unit loop a register that is reloaded from scratch in
        // front of this loop would put an s_waitcnt vmcnt(0) INTO it (seen in the ISA), i.e. one round trip per touch again
        const unsigned char* wbp = reinterpret_cast<const unsigned char*>(cv.wfrag) + (size_t)((g * MT + mt) * cv.ksteps) * 2048u;
        const int nlines = cv.ksteps * 16;
        const int ln = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
        for (int base_line = 64 * wi; base_line < nlines; base_line += 64 * WM) {
            const int line = min(base_line + ln, nlines - 1);       // (past the end: the last line again)
            RB_TOUCH(wbp + (size_t)line * 128u);
        }
    };
    if (r.warm) warm_weights(r.conv[0]);
    if constexpr (WR > 0) {
        // ring: the touches go out BEFORE the DMA requests they are meant to speed up -- a counted wait for a DMA piece also waits for
        // every older touch (loads return in order), so a touch may never sit between a piece and its wait: the weights of conv k + 2
        // are touched when conv k's loop is left, in front of the requests for conv k + 1's first groups (here: conv 1, then conv 0)
        if (r.warm) warm_weights(r.conv[1]);
    }
    for (int k = 1; k < (r.warm ? r.n_convs : 0); ++k) { // history rows [-hist_k, 0) of node k, all streams of this workgroup: 16 bytes of every 128
        const RbNode& nd = r.node[k];
        const int hk = r.conv[k].hist;
        constexpr int LPR = (C * 4 + 127) / 128;        // lines per row slice of this group
        const int total = scur * hk * LPR;
        for (int i = tid; i < total; i += NT) {
            const int s = i / (hk * LPR), rem_ = i - s * hk * LPR;
            const int rr = rem_ / LPR, li = rem_ - rr * LPR;
            int row = nd.cursor - hk + rr;
            if (row < 0) row += nd.rows;
            RB_TOUCH(nd.base + ((size_t)(b0 + s) * nd.rows + row) * nd.ch + nd.choff + g * nd.gstride + 32 * li);
        }
    }
    if constexpr (WR > 0) preload(std::integral_constant<int, TA>(), r.conv[0]);      // ring groups 0 .. WR-2 of conv 0: they land under the staging below

    // ---- stage the chain input: rows [-hist_0, T) of every stream, activated and split ----
    {
        const RbNode& nd = r.node[0];
        const int h0 = r.conv[0].hist;
        const int per = (h0 + T) * C8;                  // 8-channel pieces per stream
        const int total = scur * per;
        constexpr int SB = 4;                           // pieces per thread in flight: one memory round trip per SB * NT pieces
        for (int i0 = tid; i0 < total; i0 += SB * NT) {
            float4 u[SB], v[SB]; int dst[SB];
#pragma unroll
            for (int k = 0; k < SB; ++k) {
                const int i = i0 + k * NT;
                dst[k] = -1;
                if (i < total) {
                    const int s = i / per, rem_ = i - s * per;
                    const int rr = rem_ / C8, c8 = rem_ - rr * C8;
                    int row = nd.cursor - h0 + rr;
                    if (row < 0) row += nd.rows;
                    if (row >= nd.rows) row -= nd.rows;
                    const float4* p = reinterpret_cast<const float4*>(nd.base + ((size_t)(b0 + s) * nd.rows + row) * nd.ch + nd.choff + g * nd.gstride + 8 * c8);
                    u[k] = p[0]; v[k] = p[1];
                    dst[k] = (s * r.rps + r.hm - h0 + rr) * RS + 16 * c8;
                }
            }
#pragma unroll
            for (int k = 0; k < SB; ++k)
                if (dst[k] >= 0) rb_put8<C, ACT>(xs + dst[k], u[k], v[k], r.slope);
        }
        if constexpr (BIAS_LDS) {
            for (int i = tid; i < r.n_convs * C; i += NT) {
                const int k = i / C, c = i - k * C;
                bias_lds[i] = r.conv[k].bias ? r.conv[k].bias[g * C + c] : 0.f;
            }
        }
    }
    __syncthreads();
    RB_STAMP(1);

    bool bad = false;

    // history rows [-hist, 0) of node `k` (all streams of this workgroup): issue the loads / convert and write them in front of the new rows
    auto hist_issue = [&](int k, float4 (&hu)[NHP], float4 (&hv)[NHP]) __attribute__((always_inline)) {
        const RbNode& nd = r.node[k];
        const int hk = r.conv[k].hist;
        const int per = hk * C8, total = scur * per;
#pragma unroll
        for (int q = 0; q < NHP; ++q) {
            const int i = tid + q * NT;
            hu[q] = hv[q] = make_float4(0.f, 0.f, 0.f, 0.f);
            if (i < total) {
                const int s = i / per, rem_ = i - s * per;
                const int rr = rem_ / C8, c8 = rem_ - rr * C8;
                int row = nd.cursor - hk + rr;
                if (row < 0) row += nd.rows;
                const float4* p = reinterpret_cast<const float4*>(nd.base + ((size_t)(b0 + s) * nd.rows + row) * nd.ch + nd.choff + g * nd.gstride + 8 * c8);
                hu[q] = p[0]; hv[q] = p[1];
            }
        }
    };
    auto hist_commit = [&](int k, const float4 (&hu)[NHP], const float4 (&hv)[NHP]) __attribute__((always_inline)) {
        const int hk = r.conv[k].hist;
        const int per = hk * C8, total = scur * per;
#pragma unroll
        for (int q = 0; q < NHP; ++q) {
            const int i = tid + q * NT;
            if (i < total) {
                const int s = i / per, rem_ = i - s * per;
                const int rr = rem_ / C8, c8 = rem_ - rr * C8;
                rb_put8<C, ACT>(xs + (s * r.rps + r.hm - hk + rr) * RS + 16 * c8, hu[q], hv[q], r.slope);
            }
        }
    };
    // this wave's 16 bias values per lane: from LDS in the epilogue, or (128-channel variant) fetched under the MFMAs
    auto bias_issue = [&](int k, float4 (&breg)[4]) __attribute__((always_inline)) {
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            breg[qd] = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (!BIAS_LDS) {
                if (r.conv[k].bias) breg[qd] = *reinterpret_cast<const float4*>(r.conv[k].bias + g * C + mt * 32 + 8 * qd + 4 * lh);
            }
        }
    };
    // h = acc0 + acc1/2048 (+ bias), in the order of conv_rl16's epilogue
    auto finish = [&](int k, bool vld, const f32x16& am, const f32x16& ac, const float4 (&breg)[4], float (&h)[16]) __attribute__((always_inline)) {
        const bool hb = r.conv[k].bias != nullptr;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            float4 bb = breg[qd];
            if constexpr (BIAS_LDS) bb = *reinterpret_cast<const float4*>(bias_lds + k * C + mt * 32 + 8 * qd + 4 * lh);
            float v0 = fmaf(ac[4 * qd], kRbLoInv, am[4 * qd]), v1 = fmaf(ac[4 * qd + 1], kRbLoInv, am[4 * qd + 1]);
            float v2 = fmaf(ac[4 * qd + 2], kRbLoInv, am[4 * qd + 2]), v3 = fmaf(ac[4 * qd + 3], kRbLoInv, am[4 * qd + 3]);
            bad |= vld & (!(fabsf(v0) <= 3.0e38f) | !(fabsf(v1) <= 3.0e38f) | !(fabsf(v2) <= 3.0e38f) | !(fabsf(v3) <= 3.0e38f));
            if (hb) { v0 += bb.x; v1 += bb.y; v2 += bb.z; v3 += bb.w; }
            h[4 * qd] = v0; h[4 * qd + 1] = v1; h[4 * qd + 2] = v2; h[4 * qd + 3] = v3;
        }
    };
    // rows of node k that later calls need (or all of them): straight from registers, raw f32
    auto ring_store = [&](int k, int j, const float (&h)[16], bool all) __attribute__((always_inline)) {
        const RbNode& nd = r.node[k];
        if (!valid[j] || !(all || tj[j] >= T - nd.keep)) return;
        int row = nd.cursor + tj[j];
        if (row >= nd.rows) row -= nd.rows;
        float* p = nd.base + ((size_t)(b0 + sj[j]) * nd.rows + row) * nd.ch + nd.choff + g * nd.gstride + mt * 32 + 4 * lh;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<float4*>(p + 8 * qd) = make_float4(h[4 * qd], h[4 * qd + 1], h[4 * qd + 2], h[4 * qd + 3]);
    };
    // the residual of a unit (node k) at this lane's column j: the chain input, or what this very lane stored at the end of the previous unit
    auto res_load = [&](int k, int j, float4 (&rr)[4]) __attribute__((always_inline)) {
        const RbNode& nd = r.node[k];
        int row = nd.cursor + tj[j];
        if (row >= nd.rows) row -= nd.rows;
        const float* p = nd.base + ((size_t)(b0 + sj[j]) * nd.rows + row) * nd.ch + nd.choff + g * nd.gstride + mt * 32 + 4 * lh;
        // `nt` loads are served by L2, never by this CU's vector L1: a line of that L1 may have been filled, BEFORE this lane's
        // store, by the history fetch of the neighbouring row (rings are only 16-byte aligned, a 128-byte line can span two rows)
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const f32x4 v = __builtin_nontemporal_load(reinterpret_cast<const f32x4*>(p + 8 * qd));     // (a clamped column re-reads a valid one)
            rr[qd] = make_float4(v[0], v[1], v[2], v[3]);
        }
    };
    // act(h), split, over this lane's LDS row: the next conv's B operand
    auto lds_put = [&](int j, const float (&h)[16]) __attribute__((always_inline)) {
        if (!valid[j]) return;
        unsigned char* row = xs + lrow[j] * RS;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int ml = mt * 32 + 8 * qd + 4 * lh;
            f16x4 hi, lo;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float y = rb_act<ACT>(h[4 * qd + e], r.slope);
                const _Float16 hh = (_Float16)y;
                hi[e] = hh;
                lo[e] = (_Float16)((y - (float)hh) * kRbLoScale);
            }
            *reinterpret_cast<f16x4*>(row + 2 * ml) = hi;
            *reinterpret_cast<f16x4*>(row + 2 * C + 2 * ml) = lo;
        }
    };
    // the MFMA loop of one conv over this wave's n-tiles (a wave with one tile less than NTW runs the shorter loop)
    auto conv_loop = [&](auto taps_c, const RbConv& cv, f32x16 (&m)[NTW], f32x16 (&c)[NTW]) __attribute__((always_inline)) {
        constexpr int TAPS = decltype(taps_c)::value;
        const unsigned char* x[NTW];
#pragma unroll
        for (int j = 0; j < NTW; ++j) x[j] = xs + (lrow[j] - cv.hist) * RS + 16 * lh;
        if constexpr (WR > 0) {
            const unsigned char* gsrc = ring_src(cv);
            if (NTW > 2 && ntw < NTW) rb_mfma_ring<C, TAPS, (NTW > 2 ? NTW - 1 : NTW), NTW, WR>(x, cv.dil * RS, rg, gsrc, wave, mt, lane16, m, c);
            else rb_mfma_ring<C, TAPS, NTW, NTW, WR>(x, cv.dil * RS, rg, gsrc, wave, mt, lane16, m, c);
        } else {
            const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(cv.wfrag), 0, cv.w_bytes, 0x00020000);
            const unsigned wbase = (unsigned)((g * MT + mt) * cv.ksteps + ((ADK_RB16_DBG & 16) ? (blockIdx.x * 5) % 24 : 0)) * 2048u;
            if (NTW > 2 && ntw < NTW) rb_mfma<C, TAPS, PF, LS, (NTW > 2 ? NTW - 1 : NTW), NTW>(x, cv.dil * RS, rsrc_w, lane16, wbase, ah, al, m, c);
            else rb_mfma<C, TAPS, PF, LS, NTW, NTW>(x, cv.dil * RS, rsrc_w, lane16, wbase, ah, al, m, c);
        }
    };

    const int units = r.n_convs >> 1;
#pragma unroll 1
    for (int u = 0; u < units; ++u) {
        // ================= conv A: node 2u -> node 2u+1 =================
        {
            const int k = 2 * u;
            const RbConv cv = r.conv[k];
            f32x16 m[NTW], c[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { m[j][e] = 0.f; c[j][e] = 0.f; }
            float4 breg[4];
            if constexpr (!BIAS_LATE) bias_issue(k, breg);
            if (WR == 0 && r.warm) warm_weights(r.conv[k + 1]);
            conv_loop(std::integral_constant<int, TA>(), cv, m, c);
            RB_STAMP(2 + 4 * k);
            if constexpr (BIAS_LATE) bias_issue(k, breg);
            if (WR > 0 && r.warm && k + 2 < r.n_convs) warm_weights(r.conv[k + 2]);
            preload(std::integral_constant<int, TB>(), r.conv[k + 1]);      // the next conv's first fragments / ring groups arrive under the epilogue
            float4 hu[NHP], hv[NHP];                            // history rows of the next conv's input, in flight during the epilogue
            if constexpr (!HIST_LATE) hist_issue(k + 1, hu, hv);
            float h[NTW][16];
#pragma unroll
            for (int j = 0; j < NTW; ++j) { finish(k, valid[j], m[j], c[j], breg, h[j]); ring_store(k + 1, j, h[j], false); }
            if constexpr (HIST_LATE) { __builtin_amdgcn_sched_barrier(0); hist_issue(k + 1, hu, hv); }
            RB_STAMP(3 + 4 * k);
            __syncthreads();                            // every wave is done reading the rows of conv A's input
            RB_STAMP(4 + 4 * k);
#pragma unroll
            for (int j = 0; j < NTW; ++j) lds_put(j, h[j]);
            hist_commit(k + 1, hu, hv);
            __syncthreads();
            RB_STAMP(5 + 4 * k);
        }
        // ================= conv B: node 2u+1 -> node 2u+2, + residual (node 2u) =================
        {
            const int k = 2 * u + 1;
            const RbConv cv = r.conv[k];
            const bool last = u + 1 == units;
            const int kn = last ? k : k + 1;                    // (the last conv prefetches its own data again: harmless, keeps the code straight-line)
            f32x16 m[NTW], c[NTW];
#pragma unroll
            for (int j = 0; j < NTW; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) { m[j][e] = 0.f; c[j][e] = 0.f; }
            float4 breg[4];
            if constexpr (!BIAS_LATE) bias_issue(k, breg);
            if (WR == 0 && !last && r.warm) warm_weights(r.conv[kn]);
            // the residual (this unit's input at this lane's columns) is fetched under the MFMAs of the unit's second conv where the
            // register budget allows, else right after them
            float4 rr[NTW][4];
            if constexpr (EARLY_RES) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) res_load(k - 1, j, rr[j]);
            }
            conv_loop(std::integral_constant<int, TB>(), cv, m, c);
            RB_STAMP(2 + 4 * k);
            if constexpr (BIAS_LATE) bias_issue(k, breg);
            if constexpr (WR > 0) {
                if (!last) {                                    // (no registers involved: a branch costs nothing here)
                    if (r.warm && k + 2 < r.n_convs) warm_weights(r.conv[k + 2]);
                    preload(std::integral_constant<int, TA>(), r.conv[kn]);
                }
            } else {
                preload(std::integral_constant<int, TA>(), r.conv[kn]);
            }
            float4 hu[NHP], hv[NHP];
            if constexpr (!HIST_LATE) hist_issue(kn, hu, hv);
            if constexpr (!EARLY_RES) {
#pragma unroll
                for (int j = 0; j < NTW; ++j) res_load(k - 1, j, rr[j]);
            }
            float h[NTW][16];
#pragma unroll
            for (int j = 0; j < NTW; ++j) {
                finish(k, valid[j], m[j], c[j], breg, h[j]);
#pragma unroll
                for (int qd = 0; qd < 4; ++qd) { h[j][4 * qd] += rr[j][qd].x; h[j][4 * qd + 1] += rr[j][qd].y; h[j][4 * qd + 2] += rr[j][qd].z; h[j][4 * qd + 3] += rr[j][qd].w; }
                ring_store(k + 1, j, h[j], true);
            }
            if constexpr (HIST_LATE) { __builtin_amdgcn_sched_barrier(0); hist_issue(kn, hu, hv); }
            RB_STAMP(3 + 4 * k);
            if (!last) {
                __syncthreads();
                RB_STAMP(4 + 4 * k);
#pragma unroll
                for (int j = 0; j < NTW; ++j) lds_put(j, h[j]);
                hist_commit(kn, hu, hv);
                __syncthreads();
                RB_STAMP(5 + 4 * k);
            }
        }
    }
    if (bad) atomicOr(r.err, 8);
}

struct RbPlan { int C, ta, tb, ntw, spw, n_tiles, hm, rps, wr; size_t lds; long long blocks; };

int rb_knob(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }

int rb_streams_per_wg(int C, int batch, int t, int groups) {
    static const int env128 = rb_knob("ADK_RB16_SPW", 0);        // tuning: streams per workgroup of the 128-channel chains
    static const int env64 = rb_knob("ADK_RB16_SPW64", 0);       // ... of the 64-channel chains
    int s = (C == 128) ? 2 : 1;
    if (env128 > 0 && C == 128) s = env128;
    if (env64 > 0 && C == 64) s = env64;
    const int wm = 4 / (C / 32);                                  // waves per m-tile, up to 4 n-tiles each
    const int max_ntw = C == 128 ? 2 : 4;                         // n-tiles per wave the instantiations go up to
    while (s > 1 && (s > batch || (s * t + 31) / 32 > max_ntw * wm)) --s;
    // Round 5, few streams: a 128-channel launch of at most 128 (stream, group) items leaves half of the chip idle either way, and a workgroup's
    // time is its MFMA loops over TWO 32-column tiles -- of which one stream of a 25-step frame fills 25 columns.  One stream per workgroup and ONE
    // tile per wave (NTW = 1) halve the loops (single stream: 9.6 -> ~5 us per conv of the vocoder's second stage, profiles/r5_rb16_trace_single_stream.log)
    // at twice the workgroups.  Larger launches keep two streams per workgroup: there the chip is full and a weight fragment should feed two tiles.
    static const int few = rb_knob("ADK_RB16_FEW128", 128);
    if (C == 128 && s == 2 && env128 <= 0 && (long long)batch * groups <= few && t <= 32) s = 1;
    return s;
}

// Slots of the shared LDS weight ring for this geometry (0: the waves stream their weights into registers, as in round 3).
// On for the 168-register variants (two n-tiles per wave, three workgroups per CU): with no weight fragments in registers -- and the
// next conv's history rows requested behind the finish pass -- they spill 0-4 registers instead of 21-33 (each reload in an epilogue is
// a vmcnt(0) round trip), and a fragment crosses L2 -> CU once per workgroup instead of twice (64 channels) / four times (32).
// Measured, 256 streams (profiles/r4_ring_ab.md): vocoder stage 2 (64 ch) 176 -> 166 us, encoder block 1 64.7 -> 61.8 us, pipeline +1.5 %.
// Off for the three-tile 32-channel variant (two workgroups per CU, no spills to begin with): there the per-group barrier costs more
// than the weight stream did (vocoder stage 3 139 -> 148 us) -- ADK_RB16_RING=2 switches it on for A/B, =0 switches all rings off.
// 128 channels: every wave has an m-tile of its own, nothing to share.
// Round 5: not for launches of <= 128 workgroups (few streams: at most one workgroup on half of the CUs) -- the ring's hand-over per group (a counted
// wait + a barrier every one or two 16-k steps) costs a lone workgroup more than the second weight stream did: 1 / 32 streams 0.715 / 0.761 ms per
// step with the ring, 0.707 / 0.752 without (experiments/sessions/r5_s12.sh).
int rb_ring_slots(int C, int ntw, int spw, long long blocks) {
    static const int on = rb_knob("ADK_RB16_RING", 1);
    if (!on || (blocks <= 128 && on < 3)) return 0;
    if (C == 32 && ntw == 2) return 3;
    if (C == 32 && ntw == 3 && on >= 2) return ADK_RB16_RING_SLOTS32;
    if (C == 64 && ntw == 2 && spw == 1) return 3;        // three 4 KiB slots is what three workgroups per CU leave room for
    return 0;
}

// Geometry of the launch, or false when the chain does not fit this kernel for this call.
bool rb_plan(const ConvArgs* c, int n, RbPlan& pl) {
    const ConvArgs& a0 = c[0];
    pl.C = a0.cin_g; pl.ta = a0.taps; pl.tb = c[1].taps;
    const int T = a0.t_out;
    const int wm = 4 / (pl.C / 32);
    pl.spw = rb_streams_per_wg(pl.C, a0.batch, T, a0.groups);
    pl.n_tiles = (pl.spw * T + 31) / 32;
    pl.ntw = std::max(pl.C == 128 ? 1 : 2, (pl.n_tiles + wm - 1) / wm);            // n-tiles per wave: 2, 3 or 4 (128 channels: 1 or 2)
    if (pl.ntw > 4 || (pl.C == 128 && pl.ntw > 2)) return false;
    pl.hm = 0;
    for (int k = 0; k < n; ++k) pl.hm = std::max(pl.hm, (c[k].taps - 1) * c[k].dilation);
    if (pl.hm > kRbMaxHist) return false;
    pl.rps = pl.hm + T;
    pl.wr = rb_ring_slots(pl.C, pl.ntw, pl.spw, (long long)((a0.batch + pl.spw - 1) / pl.spw) * a0.groups);
    const bool bias_lds = pl.C < 128 && !(pl.C == 64 && (pl.spw > 1 || pl.ntw > 2 || pl.wr > 0));        // (as BIAS_LDS of the instantiation rb_by_taps picks)
    pl.lds = (size_t)pl.spw * pl.rps * (4 * pl.C + 16) + (bias_lds ? (size_t)n * pl.C * 4 : 0) + (size_t)pl.wr * 4096 + kRbTouchSink;
    if (pl.lds > 160 * 1024 || pl.spw > (pl.C == 128 ? 2 : (pl.C == 64 ? 2 : 1))) return false;      // (SMAX of the instantiations)
    pl.blocks = (long long)((a0.batch + pl.spw - 1) / pl.spw) * a0.groups;
    return pl.blocks <= 0x7fffffffLL;
}

bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// c[0..n): the convs of the chain in launch order (A_0, B_0, A_1, B_1, ...), as op_conv_args built them for this call.
bool conv_rb16_fusable(const ConvArgs* c, int n) {
    if (n < 2 || n > kRbMaxConvs || (n & 1)) return false;
    const ConvArgs& a0 = c[0];
    const int C = a0.cin_g;
    if (C != 32 && C != 64 && C != 128) return false;
    if (a0.act_in != ADK_ACT_ELU && a0.act_in != ADK_ACT_LEAKY) return false;
    const int ta = a0.taps, tb = c[1].taps;
    if (!((ta == 11 && tb == 11) || (ta == 7 && tb == 7) || (ta == 3 && tb == 3) || (ta == 7 && tb == 1))) return false;
    if (a0.act_in == ADK_ACT_ELU && !(ta == 7 && tb == 1)) return false;          // instantiated: ELU residual units, LeakyReLU residual blocks
    if (a0.act_in == ADK_ACT_LEAKY && tb == 1) return false;
    if (a0.t_out < 1 || a0.batch < 1) return false;
    for (int k = 0; k < n; ++k) {
        const ConvArgs& a = c[k];
        const bool isB = k & 1;
        if (!a.wfrag || !aligned16(a.wfrag) || (a.bias && !aligned16(a.bias))) return false;
        if (a.cin_g != C || a.cout_g != C || a.groups != a0.groups || a.stride != 1 || a.up != 1 || a.cout_real != a.groups * C) return false;
        if (a.taps != (isB ? tb : ta) || a.dilation < 1) return false;
        if (a.act_in != a0.act_in || a.slope != a0.slope || a.act_out != ADK_ACT_NONE) return false;
        if (a.batch != a0.batch || a.t_out != a0.t_out) return false;
        if ((a.in_ch % 4) || (a.in_choff % 4) || (a.in_gstride % 4) || (a.out_ch % 4) || (a.out_choff % 4) || !aligned16(a.in) || !aligned16(a.out)) return false;
        if (a.in_rows < a.t_out + (a.taps - 1) * a.dilation || a.out_rows < a.t_out) return false;
        if (k > 0) {                                    // reads exactly what the previous conv wrote, group by group
            const ConvArgs& p = c[k - 1];
            if (a.in != p.out || a.in_rows != p.out_rows || a.in_ch != p.out_ch || a.in_choff != p.out_choff || a.in_gstride != C) return false;
            if ((a.in_row0 + (a.taps - 1) * a.dilation) % a.in_rows != p.out_cursor) return false;
        }
        if (!isB) {
            if (a.res) return false;
        } else {                                        // residual = the input of the unit's first conv, at the new rows
            const ConvArgs& f = c[k - 1];
            if (!a.res || a.res != f.in || a.res_rows != f.in_rows || a.res_ch != f.in_ch || a.res_choff != f.in_choff || a.res_gstride != f.in_gstride) return false;
            if (a.res_cursor != (f.in_row0 + (f.taps - 1) * f.dilation) % f.in_rows) return false;
        }
    }
    RbPlan pl;
    return rb_plan(c, n, pl);
}

namespace {
template <int C, int ACT, int TA, int TB, int NTW, int SMAX, int WPS, int PF, bool EARLY_RES, int LS, int WR>
int rb_go(const RbArgs& r, const RbPlan& pl, hipStream_t s) {
    auto kern = conv_rb16_kernel<C, ACT, TA, TB, NTW, SMAX, WPS, PF, EARLY_RES, LS, WR>;
    if (pl.lds > 64 * 1024) {
        static bool attr_set_dev[kMaxDevices] = {};     // function attributes are per device
        bool& attr_set = attr_set_dev[current_device()];
        if (!attr_set) {
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
    }
    const unsigned grid = r.helpers > 0 ? (unsigned)(((r.n_main + 7) & ~7) + 8 * r.helpers) : (unsigned)pl.blocks;
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), pl.lds, s, r);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

template <int C, int NTW, int SMAX, int WPS, int PF, bool EARLY_RES, int LS, int WR = 0>
int rb_by_taps(const RbArgs& r, const RbPlan& pl, int act, hipStream_t s) {
    if (act == ADK_ACT_ELU) return rb_go<C, ADK_ACT_ELU, 7, 1, NTW, SMAX, WPS, PF, EARLY_RES, LS, WR>(r, pl, s);
    if (pl.ta == 11) return rb_go<C, ADK_ACT_LEAKY, 11, 11, NTW, SMAX, WPS, PF, EARLY_RES, LS, WR>(r, pl, s);
    if (pl.ta == 7) return rb_go<C, ADK_ACT_LEAKY, 7, 7, NTW, SMAX, WPS, PF, EARLY_RES, LS, WR>(r, pl, s);
    return rb_go<C, ADK_ACT_LEAKY, 3, 3, NTW, SMAX, WPS, PF, EARLY_RES, LS, WR>(r, pl, s);
}
}  // namespace

// keep[k]: rows of history the ring behind node k+1 (the output of conv k) must hold for later calls (its ring's `hist`), k < n-1
int launch_conv_rb16(const ConvArgs* c, int n, const int* keep, hipStream_t s) {
    if (!conv_rb16_fusable(c, n)) return ADK_ERR_STATE;
    RbPlan pl;
    if (!rb_plan(c, n, pl)) return ADK_ERR_STATE;
    if (c[0].n_total == 0) return ADK_OK;
    RbArgs r;
    memset(&r, 0, sizeof(r));
    const ConvArgs& a0 = c[0];
    r.node[0].base = const_cast<float*>(a0.in); r.node[0].rows = a0.in_rows; r.node[0].ch = a0.in_ch;
    r.node[0].cursor = (a0.in_row0 + (a0.taps - 1) * a0.dilation) % a0.in_rows;
    r.node[0].choff = a0.in_choff; r.node[0].gstride = a0.in_gstride; r.node[0].keep = 0;
    for (int k = 0; k < n; ++k) {
        const ConvArgs& a = c[k];
        RbNode& nd = r.node[k + 1];
        nd.base = a.out; nd.rows = a.out_rows; nd.ch = a.out_ch; nd.cursor = a.out_cursor; nd.choff = a.out_choff; nd.gstride = a.cout_g;
        nd.keep = k + 1 < n ? std::max(keep[k], (c[k + 1].taps - 1) * c[k + 1].dilation) : 0;
        RbConv& cv = r.conv[k];
        cv.wfrag = a.wfrag; cv.bias = a.bias; cv.dil = a.dilation; cv.hist = (a.taps - 1) * a.dilation;
        cv.ksteps = (a.ktot + 63) / 64 * 4;
        cv.w_bytes = (unsigned)((unsigned long long)a.groups * (a.cout_g / 32) * cv.ksteps * 2048ull);
    }
    r.n_convs = n; r.batch = a0.batch; r.t = a0.t_out; r.groups = a0.groups;
    r.spw = pl.spw; r.hm = pl.hm; r.rps = pl.rps; r.n_tiles = pl.n_tiles;
    r.slope = a0.slope; r.err = conv_err_word(a0);
    // L2 warm-up where a conv's weight block is a few loads per lane (32 / 64 channels) AND the launch is one workgroup per CU or
    // less (the encoder's chains: 256 workgroups, every round trip exposed -- encoder block 1 64 -> 68 us without it).  A touch is one
    // dword per 128-byte line, i.e. 64 cache lines per wave instruction: with three workgroups per CU starting at once the ~27 touch
    // instructions per wave in front of the staging loads cost more than they save -- round 4, after the pinned prefetch and the
    // weight ring: vocoder stage 2 (768 workgroups) stage-in 16.7 us, chain 168 -> 152 us without the touches, stage 3 139 -> 137.
    // The 128-channel chains would spend 22 line touches per lane and conv on it (17 us of prologue, no gain in the loops:
    // profiles/r3_rb16_timeline.md) -- their waves walk the weights in lockstep instead (LS).
    // ADK_RB16_WARM: 0 never, 1 always, 2 = the round-3 rule (32 / 64 channels, any launch size).
    static const int warm_env = rb_knob("ADK_RB16_WARM", -1);
    // Round 5: not for launches of <= 128 workgroups either (few streams) -- there the touches cost more than they save: 1 / 8 / 32 streams
    // 0.719 / 0.750 / 0.772 ms per step with them, 0.713 / 0.746 / 0.761 without (experiments/sessions/r5_s11.sh); with them on the 128-channel chains too: 0.760 / 0.824 / 0.818.
    r.warm = warm_env == 2 ? (pl.C <= 64) : (warm_env >= 0 ? warm_env : (pl.C <= 64 && pl.blocks <= 384 && pl.blocks > 128));
#if ADK_RB16_DBG & 1
    r.dbg_slot = g_rb_launch++;
#endif
    {
        // helper workgroups (see the kernel head): launches of at most kHelperBlocks computing workgroups; per XCD the weights of the groups it hosts
        // must fit its L2 beside everything else: the longest prefix of convs within kHelperBytes.  ADK_RB16_HELPERS: helpers per XCD (0 = off).
        static const int per_xcd = rb_knob("ADK_RB16_HELPERS", 4);
        static const int max_blocks = rb_knob("ADK_RB16_HELPER_BLOCKS", 64);
        constexpr long long kHelperBytes = 3584 * 1024;
        r.n_main = (int)pl.blocks; r.helpers = 0; r.helper_convs = 0;
        if (per_xcd > 0 && pl.blocks <= max_blocks && a0.groups <= 32) {
            int worst = 0;                                   // most distinct groups on one XCD
            for (int x = 0; x < 8; ++x) {
                unsigned seen = 0; int cnt = 0;
                for (long long i = x; i < pl.blocks; i += 8) { const unsigned bit = 1u << (int)(i % a0.groups); if (!(seen & bit)) { seen |= bit; ++cnt; } }
                worst = std::max(worst, cnt);
            }
            long long bytes = 0;
            int kc = 0;
            for (; kc < n; ++kc) {
                const long long one = (long long)(pl.C / 32) * r.conv[kc].ksteps * 2048ll * worst;
                if (bytes + one > kHelperBytes) break;
                bytes += one;
            }
            if (kc > 0 && worst > 0) { r.helpers = per_xcd; r.helper_convs = kc; }
        }
    }
    const int act = a0.act_in;
    // Register budgets.  Two n-tiles per wave: 168 registers (three 4-wave workgroups per CU), the residual fetched under the second
    // conv's MFMAs; three: 256 (two workgroups), residual early; four: 256, residual after the loop.  128 channels: 256, prefetch
    // distance 3 steps (ADK_RB16_PF128), and the waves re-align every 4 steps (LS).
    // (32 channels used to run as 5 waves x 2 tiles: the hardware starts every workgroup's waves on the same SIMD, so one SIMD
    // carried two waves of every co-resident workgroup and a second 5-wave workgroup did not even fit at 168 registers --
    // profiles/r3_rb16_timeline.md.  Now 4 waves x (3, 3, 2, 2) tiles, the odd tiles rotating with the workgroup.)
    // Round 4: with the shared LDS weight ring (pl.wr slots, rb_mfma_ring) no weight fragment lives in registers -- the PF argument is
    // then unused (1) -- and the 168-register variants have room for the residual again.
    if (pl.C == 32 && pl.wr > 0) {
        if (pl.ntw == 2) return rb_by_taps<32, 2, 1, 3, 1, true, 0, 3>(r, pl, act, s);
        if (pl.ta == 11) return rb_by_taps<32, 3, 1, 2, 1, false, 0, ADK_RB16_RING_SLOTS32>(r, pl, act, s);
        return rb_by_taps<32, 3, 1, 2, 1, true, 0, ADK_RB16_RING_SLOTS32>(r, pl, act, s);
    }
    if (pl.C == 64 && pl.wr > 0) return rb_by_taps<64, 2, 1, 3, 1, ADK_RB16_EARLY64, 0, 3>(r, pl, act, s);
    // (launches of <= 128 workgroups, round 5: no ring -- and no reason to squeeze the two-tile variants into 168 registers for a third workgroup per CU
    // that never comes: the 256-register builds of the same code have no spills, the 168-register ones 30-33 spilled registers without the ring)
    const bool few = pl.blocks <= 128;
    if (pl.C == 32) {
        if (pl.ntw == 2 && few) return rb_by_taps<32, 2, 1, 2, ADK_RB16_PF32, true, 0>(r, pl, act, s);
        if (pl.ntw == 2) return rb_by_taps<32, 2, 1, 3, ADK_RB16_PF32, true, 0>(r, pl, act, s);
        // (three tiles, K11 blocks of the vocoder: with the residual fetched AFTER the second conv's loop the kernel has 1 spilled register
        // instead of 45 -- stage-3 chain 151.6 -> 142.5 us, pipeline +1.2 % on one box; the K7 + 1x1 units of the encoder have no spills
        // either way and lose 2 us with the late fetch)
        if (pl.ntw == 3 && pl.ta == 11) return rb_by_taps<32, 3, 1, 2, ADK_RB16_PF32, false, 0>(r, pl, act, s);
        if (pl.ntw == 3) return rb_by_taps<32, 3, 1, 2, ADK_RB16_PF32, true, 0>(r, pl, act, s);
        return rb_by_taps<32, 4, 1, 2, 2, false, 0>(r, pl, act, s);
    }
    if (pl.C == 64) {
        if (pl.ntw == 2 && pl.spw == 1 && few) return rb_by_taps<64, 2, 1, 2, ADK_RB16_PF64, true, 0>(r, pl, act, s);
        if (pl.ntw == 2 && pl.spw == 1) return rb_by_taps<64, 2, 1, 3, ADK_RB16_PF64, true, 0>(r, pl, act, s);
        if (pl.ntw <= 3) return rb_by_taps<64, 3, 2, 2, 2, true, 0>(r, pl, act, s);
        return rb_by_taps<64, 4, 2, 2, 2, false, 0>(r, pl, act, s);
    }
    // few streams: one stream, one tile per wave.  (Deeper weight prefetch for these one-workgroup-per-CU launches -- 8 steps at 128 channels, 4 at 32, five
    // ring slots at 64 -- measured 1-2 % SLOWER at 1 / 8 / 32 / 64 streams, round 5: the stream of one workgroup per CU is not bound by its round trips.)
    if (pl.ntw == 1) return rb_by_taps<128, 1, 1, 2, ADK_RB16_PF128, true, 4>(r, pl, act, s);
    return rb_by_taps<128, 2, 2, 2, ADK_RB16_PF128, true, 4>(r, pl, act, s);
}

const char* conv_rb16_name(const ConvArgs* c, int n) {
    (void)n;
    return c[0].cin_g == 32 ? "conv_rb16<32>" : (c[0].cin_g == 64 ? "conv_rb16<64>" : "conv_rb16<128>");
}

}  // namespace adk
