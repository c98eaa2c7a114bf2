// NOT COMPILED -- kept for the record.  The four-wave LDS-DMA form of conv_ou16 (round 6, first form): one wave per SIMD, 32-column tiles on
// v_mfma_f32_32x32x16_f16; 13.3 us per 256-stream launch by rocprofv3 (profiles/r6_kernel_stats.csv of commit 0b4e... era captures:
// profiles/r6_ou16_timeline.md has the phase clocks).  Replaced by the eight-wave form (audiodec_amd/csrc/conv_ou16.hip) which runs the same
// schedule at two waves per SIMD: 8.9 against 10.2 us median workgroup time at 256 streams.  To try it again: paste the kernel below into
// conv_ou16.hip and launch it with 256 threads and LDS = W1 + W2 + 4 * OU_RING + 4 * OU_RSC + 16 + 4 KiB + 256 B.
//
// conv_ou16 -- the hand-over between two up-sampling stages of the HiFi-GAN vocoder as ONE streaming kernel:
//     c = conv_out(x)                      MultiGroupConv1d.inference, 1x1 conv over the groups   models/vocoder/modules/multi_fusion.py:139-141
//     c = upsamples[i+1].inference(act(c)) LeakyReLU -> ConvTranspose1d + bias                    models/vocoder/HiFiGAN.py:285-289
// for the last stage boundary (192 -> 64 channels, then 64 -> 32, K 6, stride 3): the fused decoder ConvTranspose1d + activation
// kernel of the north-star, with the 1x1 conv that feeds it pulled in, so that the 64-channel tensor between them
// (25.6 KB per stream and frame, written and read back by the two-launch form) never exists in memory.
//
// Per stream and frame: reads 100 rows x 768 B, writes 300 rows x 128 B -- 115 KB for 2 x 2.46 MFLOP; 29.5 MB per 256-stream launch.
// One workgroup = one stream, wave w = time tile w (32 steps); every byte arrives by LDS-DMA (round 6; the register-fed form of round 3 is
// experiments/conv_ou16_round3_register_fed.hip):
//   * both weight sets (2 x 48 KB of split-f16 fragments) and the wave's 32 rows x 768 B of activations -- as six column blocks through a
//     wave-private ring of three 4-KiB slots, XOR-swizzled through the DMA's source addresses -- see the kernel's own comment below;
//   * GEMM 1 (64 x 192 per step) leaves c in the accumulators; LeakyReLU(c), split into f16 hi / lo, overwrites the wave's ring (row r of the
//     tile at r * 272 B); the row in front of a wave's first step comes from the previous wave (halo rows; wave 0: the history row of the
//     transposed conv, the last c of the previous call, from the state ring);
//   * GEMM 2 (polyphase transposed conv: 96 rows x 2 taps x 64 channels) reads its B fragments from there, tap 0 one row up; every m-tile
//     has its own accumulators and the finish is branch-free, so the stores of m-tile i issue beneath the MFMAs of m-tile i + 1;
//   * outputs leave as 16-byte buffer stores; of c only the LAST row goes to its ring (the next call's history), behind everything else.
// The operations per output element and their order are those of the two separate kernels (conv_sk16 for the 1x1 conv with
// K = 192 = three unsplit chunks, conv_up16 for the transposed conv): the result is bit-identical.
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

#ifndef ADK_OU16_DBG
#define ADK_OU16_DBG 0      // tuning builds only: 1 = per-workgroup wall-clock stamps (s_memrealtime, 100 MHz) of wave 0
#endif

namespace adk {

#if ADK_OU16_DBG & 1
// [workgroup 0..1023][stamp]: 0 entry, 1 the 37 LDS-DMA instructions of the wave issued, 2 first barrier passed ({biases, W1, column block 0} landed),
// 3 GEMM 1 done (72 MFMAs; column blocks 1-5 streamed in beneath it), 4 act(c) written to LDS + second barrier passed, 5 GEMM 2 MFMAs and stores issued (exit)
__device__ unsigned long long g_ou_trace[1024 * 8];
extern "C" int adk_debug_ou_trace(unsigned long long* out, int n) {
    if (n > 1024 * 8) n = 1024 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_ou_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#define OU_STAMP(i) do { __builtin_amdgcn_sched_barrier(0); if (wave == 0 && blockIdx.x < 1024) { const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); if (lane == 0) g_ou_trace[blockIdx.x * 8 + (i)] = t_; } __builtin_amdgcn_sched_barrier(0); } while (0)
#else
#define OU_STAMP(i) do { } while (0)
#endif

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8u __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4u __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr float kOuLoScale = 2048.f, kOuLoInv = 1.f / 2048.f;
constexpr int OU_CM = 64;                        // channels between the two convs
constexpr int OU_KS2 = 2 * OU_CM / 16;           // 2 taps x 64 channels = 8 chunks of 16
constexpr int OU_RSC = 4 * OU_CM + 16;           // LDS row stride of the c buffer: [64 halfs hi][64 halfs lo][16 B pad]
constexpr int OU_TMAX = 128;                     // steps of the 1x1 conv per workgroup (4 waves x 32)

struct OuArgs { int ks1p; float inv_cout_real; int* err; unsigned out_bytes; };
typedef unsigned u32x4o __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ float ou_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

// ================================================================================================
// Round 6: the launch fed by LDS-DMA (conv_ou16_dma_kernel).
//
// What the timeline of the register-fed kernel of round 3 said (profiles/r3_ou16_timeline.md, 256 streams): 4.4 us of its 12.5 us are the ISSUE of its 48
// vector-memory instructions per wave -- every lane loads the B fragments of ITS time step, i.e. one wave instruction touches 32 ring rows
// and uses 32 bytes of each: four times the cache lines a coalesced access needs, all of it before the first MFMA; another 1.9 us ("epilogue
// 1") were the wait for ONE 256-byte store (the next call's history row of c) behind the s_waitcnt vmcnt(0) that W2's arrival needs.  Here:
//   * the activations come in by LDS-DMA too, fully coalesced: the 32 rows x 768 B of a wave's time tile are walked as six column blocks of
//     32 channels (32 rows x 128 B = four 1-KiB DMA instructions: 8 whole cache lines each), through a wave-private ring of three 4-KiB
//     slots -- the first three blocks are requested with the weights, block i + 3 when block i has been read into registers, so GEMM 1
//     starts as soon as W1 and the first block have landed and the rest of the tile streams in beneath its MFMAs.  The 16-byte pieces of a
//     row are XOR-swizzled by (row >> 1) & 7 through the per-lane SOURCE address (LDS-DMA writes lane-linear): the lanes of a ds_read_b128
//     group then hit 16 different 16-byte bank slots (cdna_hip_programming.md rule 21);
//   * no lane holds staged activations in registers (96 fewer), no ordinary vector load is left in the kernel -- biases and the history row
//     of c arrive through one more DMA instruction -- so every s_waitcnt vmcnt is written here, counted, and none drains W2 early;
//   * the history row of c for the next call goes to LDS in epilogue 1 and to memory behind GEMM 2's stores.
// Per output element the operations and their order are those of the round-3 kernel, i.e. of conv_sk16 + conv_up16: bit-identical.
#define OU_DMA16(gptr, m0val) do { unsigned m0_keep_; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                                                            : "=&s"(m0_keep_) : "v"(gptr), "s"(m0val) : "memory"); } while (0)      /* M0 is the compiler's: put back */
constexpr int OU_SLOT = 4096;                    // one column block of a wave's tile: 32 rows x 128 B
constexpr int OU_RING = 3 * OU_SLOT;             // ring of a wave; after GEMM 1 its first 32 * OU_RSC bytes hold the wave's rows of act(c)
constexpr int OU_NCB = 6;                        // column blocks of the 1x1 conv's input: 192 channels / 32

template <int N> __device__ __forceinline__ void ou_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 12) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (N == 16) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
    else if constexpr (N == 20) asm volatile("s_waitcnt vmcnt(20)" ::: "memory");
    else static_assert(N == 0, "add the count");
}

template <int ACT, int MT2>
__global__ __launch_bounds__(256, 1) void conv_ou16_dma_kernel(ConvArgs a1, ConvArgs a2, OuArgs u) {
    constexpr int KS1 = 2 * OU_NCB;
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x;
    const int T = a1.t_out;
    const int t = wave * 32 + l31;
    const bool valid = t < T;
    OU_STAMP(0);

    constexpr int W1B = 2 * KS1 * 2048, W2B = MT2 * OU_KS2 * 2048;
    unsigned char* w1l = lds;                              // [2 m-tiles][12 chunks][hi | lo][64 lanes][16 B]
    unsigned char* w2l = lds + W1B;                        // [MT2][8 chunks][hi | lo][64 lanes][16 B]
    unsigned char* ring = w2l + W2B + wave * OU_RING;      // this wave's ring; later its rows of act(c): row r at r * OU_RSC
    unsigned char* halo = w2l + W2B + 4 * OU_RING;         // [4][OU_RSC]: act(c) of the step in front of wave w's first one (w = 0: the history row)
    unsigned char* stage = halo + 4 * OU_RSC + 16 + wave * 1024;   // [4][1 KiB]: every wave fetches {c[-1] | bias 2 | bias 1}; wave 0's copy is the one used
    unsigned char* stage0 = halo + 4 * OU_RSC + 16;
    float* clast = reinterpret_cast<float*>(stage0 + 4 * 1024);    // [64]: the last step's c (+ bias): the next call's history row
    typedef unsigned char __attribute__((address_space(3)))* lds_u8_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_u8_t)lds;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- every byte of the launch by LDS-DMA, oldest first: {c[-1], biases}, W1, column blocks 0-2 of this wave's tile, W2 ----
    {
        // one instruction: lanes 0-15 the 256 B of c[-1] (what the previous call left in front of the cursor of the 64-channel ring), lanes 16-39 bias 2
        // (32 * MT2 floats), lanes 40-55 bias 1 (64 floats); lanes without a source read the head of W1 (valid memory, never used)
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a1.wfrag) + lane16;
        if (lane < 16) src = reinterpret_cast<const unsigned char*>(a2.in + ((size_t)b * a2.in_rows + a2.in_row0) * a2.in_ch + a2.in_choff) + lane16;
        else if (lane < 16 + 8 * MT2) { if (a2.bias) src = reinterpret_cast<const unsigned char*>(a2.bias) + (lane - 16) * 16; }
        else if (lane >= 40 && lane < 56) { if (a1.bias) src = reinterpret_cast<const unsigned char*>(a1.bias) + (lane - 40) * 16; }
        OU_DMA16(src, lds0 + (unsigned)(stage - lds));
    }
    {
        const unsigned char* g1 = reinterpret_cast<const unsigned char*>(a1.wfrag) + (size_t)tid * 16;
        const unsigned l1 = lds0 + (unsigned)wave * 1024u;             // wave-uniform base; the lane offset is implicit
#pragma unroll
        for (int i = 0; i < W1B / 4096; ++i) OU_DMA16(g1 + 4096 * i, l1 + 4096u * i);
    }
    // this lane's source of piece (instruction j, column block cb): row r = 8 j + lane / 8 of the tile, 16-byte piece (lane & 7) ^ ((r >> 1) & 7) of the block
    const unsigned char* xsrc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int r = 8 * j + (lane >> 3);
        int tr = wave * 32 + r;
        if (tr >= T) tr = T - 1;                            // rows past the end: a copy of the last one; nothing of theirs is stored
        int row = a1.in_row0 + tr;
        if (row >= a1.in_rows) row -= a1.in_rows;
        xsrc[j] = reinterpret_cast<const unsigned char*>(a1.in + ((size_t)b * a1.in_rows + row) * a1.in_ch + a1.in_choff) + 16 * ((lane & 7) ^ ((r >> 1) & 7));
    }
    const unsigned ring0 = lds0 + (unsigned)(ring - lds);
    auto issue_block = [&](int cb) __attribute__((always_inline)) {
        const unsigned dst = ring0 + (unsigned)(cb % 3) * OU_SLOT;
#pragma unroll
        for (int j = 0; j < 4; ++j) OU_DMA16(xsrc[j] + 128 * cb, dst + 1024u * j);
    };
    issue_block(0); issue_block(1); issue_block(2);
    {
        const unsigned char* g2 = reinterpret_cast<const unsigned char*>(a2.wfrag) + (size_t)tid * 16;
        const unsigned l2 = lds0 + (unsigned)W1B + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < W2B / 4096; ++i) OU_DMA16(g2 + 4096 * i, l2 + 4096u * i);
    }
    OU_STAMP(1);
    // in flight per wave: 1 + 12 + 12 + 4 * MT2.  GEMM 1 may start when {stage, W1, block 0} have landed: blocks 1-2 and W2 stay in flight.
    // (Issuing only {stage, W1, block 0} first and the rest beneath GEMM 1 was measured: first MFMA at 2.1 instead of 3.2 us, GEMM 1 4.1
    // instead of 2.8 us -- the waves then stall at the DMA issue inside the loop; the CU's fill rate is the limit either way:
    // profiles/r6_ou16_timeline.md)
    ou_wait_vm<8 + 4 * MT2>();
    __syncthreads();                                       // W1 is the four waves' copies
    OU_STAMP(2);
    if (tid < OU_CM / 4) {                                 // history row: activation, split, into the halo row of wave 0
        const float4 hrow = *reinterpret_cast<const float4*>(stage0 + 16 * tid);
        const float x[4] = {hrow.x, hrow.y, hrow.z, hrow.w};
        f16x4u hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = ou_act<ACT>(x[e], a2.slope);
            const _Float16 h = (_Float16)v;
            hi[e] = h; lo[e] = (_Float16)((v - (float)h) * kOuLoScale);
        }
        *reinterpret_cast<f16x4u*>(halo + 8 * tid) = hi;
        *reinterpret_cast<f16x4u*>(halo + 2 * OU_CM + 8 * tid) = lo;
    }
    const float* b2l = reinterpret_cast<const float*>(stage0 + 256);
    const float* b1l = reinterpret_cast<const float*>(stage0 + 640);

    float chk = 0.f;                                        // stays 0 while every output is finite (columns past the end are copies of the last step)
    // ---- GEMM 1: c[m][t] = sum_k W1[m][k] x[k][t], 64 rows (two m-tiles) x this wave's 32 steps; k walks the column blocks ----
    {
        f32x16 am[2], ac[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { am[mt][e] = 0.f; ac[mt][e] = 0.f; }
        const unsigned swz = (unsigned)((l31 >> 1) & 7);
        // A fragments one 16-k step ahead, in two register sets (W1 is resident: the reads do not depend on the column blocks' arrival)
        f16x8u Ah[2][2], Al[2][2];
        auto load_a = [&](int s_, int set) __attribute__((always_inline)) {
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned char* wp = w1l + (size_t)(mt * KS1 + s_) * 2048 + lane * 16;
                Ah[set][mt] = *reinterpret_cast<const f16x8u*>(wp);
                Al[set][mt] = *reinterpret_cast<const f16x8u*>(wp + 1024);
            }
        };
        load_a(0, 0);
#pragma unroll
        for (int cb = 0; cb < OU_NCB; ++cb) {
            // block cb has landed (loads return in order: what was issued behind it may stay in flight)
            if (cb == 1 || cb == 2) ou_wait_vm<8 + 4 * MT2>();
            else if (cb == 3) ou_wait_vm<8>();
            else if (cb == 4) ou_wait_vm<4>();
            else if (cb == 5) ou_wait_vm<0>();
            const unsigned char* xs = ring + (cb % 3) * OU_SLOT + l31 * 128;
            float4 xr[2][2];
#pragma unroll
            for (int sc = 0; sc < 2; ++sc)
#pragma unroll
                for (int h = 0; h < 2; ++h) xr[sc][h] = *reinterpret_cast<const float4*>(xs + 16 * ((unsigned)(4 * sc + 2 * lh + h) ^ swz));
            if (cb + 3 < OU_NCB) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads of this slot have returned: it may be overwritten
                issue_block(cb + 3);
            }
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                const int s = 2 * cb + sc;
                if (s + 1 < KS1) load_a(s + 1, (s + 1) & 1);
                const float x[8] = {xr[sc][0].x, xr[sc][0].y, xr[sc][0].z, xr[sc][0].w, xr[sc][1].x, xr[sc][1].y, xr[sc][1].z, xr[sc][1].w};
                f16x8u bh, bl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 h = (_Float16)x[e];
                    bh[e] = h; bl[e] = (_Float16)((x[e] - (float)h) * kOuLoScale);
                }
                // per accumulator the sequence of the register-fed kernel (hi*hi | hi*lo, lo*hi); the two m-tiles interleaved so that no MFMA
                // follows the one it depends on
                am[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1][0], bh, am[0], 0, 0, 0);
                am[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1][1], bh, am[1], 0, 0, 0);
                ac[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1][0], bl, ac[0], 0, 0, 0);
                ac[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah[s & 1][1], bl, ac[1], 0, 0, 0);
                ac[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s & 1][0], bh, ac[0], 0, 0, 0);
                ac[1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al[s & 1][1], bh, ac[1], 0, 0, 0);
            }
        }
        OU_STAMP(3);
        // c (+ bias): act(c), split, over this wave's ring (every block of it has been read); the last step's raw row to `clast`; the row
        // the NEXT wave's first step needs as its older tap to that wave's halo row
        unsigned char* lrow = ring + l31 * OU_RSC;
        unsigned char* hrow2 = halo + (wave + 1) * OU_RSC;
        const bool is_last = valid && t == T - 1;
        const bool to_halo = l31 == 31 && wave < 3;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = mt * 32 + 8 * qd + 4 * lh;
                float v[4] = {fmaf(ac[mt][4 * qd], kOuLoInv, am[mt][4 * qd]), fmaf(ac[mt][4 * qd + 1], kOuLoInv, am[mt][4 * qd + 1]),
                              fmaf(ac[mt][4 * qd + 2], kOuLoInv, am[mt][4 * qd + 2]), fmaf(ac[mt][4 * qd + 3], kOuLoInv, am[mt][4 * qd + 3])};
                chk = fmaf(v[0], 0.f, chk); chk = fmaf(v[1], 0.f, chk); chk = fmaf(v[2], 0.f, chk); chk = fmaf(v[3], 0.f, chk);     // inf * 0 = NaN, NaN stays: chk == 0 <=> all finite
                if (a1.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(b1l + ml);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (is_last) *reinterpret_cast<float4*>(clast + ml) = make_float4(v[0], v[1], v[2], v[3]);
                f16x4u hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float y = ou_act<ACT>(v[e], a2.slope);
                    const _Float16 h = (_Float16)y;
                    hi[e] = h; lo[e] = (_Float16)((y - (float)h) * kOuLoScale);
                }
                *reinterpret_cast<f16x4u*>(lrow + 2 * ml) = hi;            // (columns past the end hold a copy of the last step: finite, never stored)
                *reinterpret_cast<f16x4u*>(lrow + 2 * OU_CM + 2 * ml) = lo;
                if (to_halo) {
                    *reinterpret_cast<f16x4u*>(hrow2 + 2 * ml) = hi;
                    *reinterpret_cast<f16x4u*>(hrow2 + 2 * OU_CM + 2 * ml) = lo;
                }
            }
    }
    __syncthreads();                                       // (W2 landed with the last column block: vmcnt(0) above)
    OU_STAMP(4);

    // ---- GEMM 2: the polyphase transposed conv; k = (tap j, channel), tap 0 = the older row c[t-1], tap 1 = c[t] ----
    int orow0 = a2.out_cursor + t * a2.up;
    orow0 %= a2.out_rows;
    const unsigned char* x0 = (l31 == 0 ? halo + wave * OU_RSC : ring + (l31 - 1) * OU_RSC) + 16 * lh;
    const unsigned char* x1 = ring + l31 * OU_RSC + 16 * lh;
    f16x8u bh[OU_KS2], bl[OU_KS2];
#pragma unroll
    for (int s = 0; s < OU_KS2; ++s) {
        const unsigned char* p = (s / 4 ? x1 : x0) + 32 * (s % 4);
        bh[s] = *reinterpret_cast<const f16x8u*>(p);
        bl[s] = *reinterpret_cast<const f16x8u*>(p + 2 * OU_CM);
    }
    // Every m-tile has its own accumulators and the body below is ONE basic block: the finish of m-tile mt (accumulator reads, bias, the
    // finite check, address arithmetic, stores) is scheduled beneath the MFMAs of m-tile mt + 1 instead of holding them up -- with one pair of
    // accumulators for all m-tiles the next MFMAs waited for the reads of the previous finish, and every `if` of the finish (bias, columns
    // past the end) cut the block.  So: a missing bias is a select, columns past the end store out of bounds (raw buffer stores drop those).
    // (Measured and dropped, profiles/r6_ou16_timeline.md: A fragments requested one step ahead in two register sets -- GEMM 2 3.9 instead of 3.1 us;
    // finished rows through an LDS tile so that every store instruction writes 8 whole 128-byte rows -- all 12 stores then issue at the
    // very end instead of beneath the next m-tile's MFMAs: 5.4 us.)
    const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(a2.out, 0, u.out_bytes, 0x00020000);
    const unsigned out_base = ((unsigned)b * (unsigned)a2.out_rows * (unsigned)a2.out_ch + (unsigned)a2.out_choff) * 4u;
    const unsigned row_bytes = (unsigned)a2.out_ch * 4u;
    const bool has_b2 = a2.bias != nullptr;
    const unsigned oob_mask = valid ? 0u : 0x80000000u;    // columns past the end: the store goes out of bounds (dropped), no branch
    f32x16 am2[MT2], ac2[MT2];
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt)
#pragma unroll
        for (int e = 0; e < 16; ++e) { am2[mt][e] = 0.f; ac2[mt][e] = 0.f; }
#pragma unroll
    for (int mt = 0; mt <= MT2; ++mt) {
        if (mt < MT2) {
            const unsigned char* wp = w2l + (size_t)mt * OU_KS2 * 2048 + lane * 16;
#pragma unroll
            for (int s = 0; s < OU_KS2; ++s) {
                const f16x8u Ah = *reinterpret_cast<const f16x8u*>(wp + s * 2048);
                const f16x8u Al = *reinterpret_cast<const f16x8u*>(wp + s * 2048 + 1024);
                am2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh[s], am2[mt], 0, 0, 0);
                ac2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl[s], ac2[mt], 0, 0, 0);
                ac2[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh[s], ac2[mt], 0, 0, 0);
            }
        }
        if (mt > 0) {
            const int m1 = mt - 1;                         // finish of the previous m-tile
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = m1 * 32 + 8 * qd + 4 * lh;   // GEMM row = phase * cout_real + co
                float4 v = make_float4(fmaf(ac2[m1][4 * qd], kOuLoInv, am2[m1][4 * qd]), fmaf(ac2[m1][4 * qd + 1], kOuLoInv, am2[m1][4 * qd + 1]),
                                       fmaf(ac2[m1][4 * qd + 2], kOuLoInv, am2[m1][4 * qd + 2]), fmaf(ac2[m1][4 * qd + 3], kOuLoInv, am2[m1][4 * qd + 3]));
                chk = fmaf(v.x, 0.f, chk); chk = fmaf(v.y, 0.f, chk); chk = fmaf(v.z, 0.f, chk); chk = fmaf(v.w, 0.f, chk);
                float4 bb = *reinterpret_cast<const float4*>(b2l + ml);
                bb.x = has_b2 ? bb.x : 0.f; bb.y = has_b2 ? bb.y : 0.f; bb.z = has_b2 ? bb.z : 0.f; bb.w = has_b2 ? bb.w : 0.f;
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                const int ph = (int)(((float)ml + 0.5f) * u.inv_cout_real);        // ml / cout_real, exact for these sizes
                int r2 = orow0 + ph;
                if (r2 >= a2.out_rows) r2 -= a2.out_rows;
                const unsigned off = (out_base + (unsigned)r2 * row_bytes + (unsigned)(ml - ph * a2.cout_real) * 4u) | oob_mask;      // (offsets stay below 2^31: checked by the launcher)
                u32x4o pv;
                pv.x = __float_as_uint(v.x); pv.y = __float_as_uint(v.y); pv.z = __float_as_uint(v.z); pv.w = __float_as_uint(v.w);
                __builtin_amdgcn_raw_buffer_store_b128(pv, rsrc_out, off, 0, 0);
            }
        }
    }
    if (tid < OU_CM / 4) {                                  // the next call's history row of c, behind everything else
        int row = a1.out_cursor + T - 1;
        if (row >= a1.out_rows) row -= a1.out_rows;
        *reinterpret_cast<float4*>(a1.out + ((size_t)b * a1.out_rows + row) * a1.out_ch + a1.out_choff + 4 * tid) = *reinterpret_cast<const float4*>(clast + 4 * tid);
    }
    OU_STAMP(5);
    if (!(chk == 0.f)) atomicOr(u.err, 8);
}

}  // namespace
}  // namespace adk
