// conv_ou16 -- the hand-over between two up-sampling stages of the HiFi-GAN vocoder as ONE streaming kernel:
//     c = conv_out(x)                      MultiGroupConv1d.inference, 1x1 conv over the groups   models/vocoder/modules/multi_fusion.py:139-141
//     c = upsamples[i+1].inference(act(c)) LeakyReLU -> ConvTranspose1d + bias                    models/vocoder/HiFiGAN.py:285-289
// for the last stage boundary (192 -> 64 channels, then 64 -> 32, K 6, stride 3): the fused decoder ConvTranspose1d + activation
// kernel of the north-star, with the 1x1 conv that feeds it pulled in, so that the 64-channel tensor between them
// (25.6 KB per stream and frame, written and read back by the two-launch form) never exists in memory.
//
// Per stream and frame: reads 100 rows x 768 B, writes 300 rows x 128 B -- 115 KB for 2 x 2.46 MFLOP; 29.5 MB per 256-stream launch.
// One workgroup = one stream, EIGHT waves (two per SIMD), wave w = time tile w (16 steps); every byte arrives by LDS-DMA:
//   * both weight sets (2 x 48 KB of split-f16 fragments) and the wave's 16 rows x 768 B of activations -- as six column blocks through a
//     wave-private ring of three 2-KiB slots, XOR-swizzled through the DMA's source addresses -- see the kernel's own comment below;
//   * GEMM 1 (64 x 192 per step) leaves c in the accumulators; LeakyReLU(c), split into f16 hi / lo, overwrites the wave's ring (row r of the
//     tile at r * 288 B); the row in front of a wave's first step comes from the previous wave (halo rows; wave 0: the history row of the
//     transposed conv, the last c of the previous call, from the state ring);
//   * GEMM 2 (polyphase transposed conv: 96 rows x 2 taps x 64 channels) reads its B fragments from there, tap 0 one row up; every m-tile
//     has its own accumulators and the finish is branch-free, so the stores of m-tile i issue beneath the MFMAs of m-tile i + 1;
//   * outputs leave as 16-byte buffer stores; of c only the LAST row goes to its ring (the next call's history), behind everything else.
// The products per output element and their chunk order are those of the two separate kernels (conv_sk16 for the 1x1 conv, conv_up16 for the
// transposed conv); the 16 x 16 x 32 MFMA groups the f32 additions differently from their 32 x 32 x 16: equal to f32 round-off, not bit for bit.
// Earlier forms, not compiled: experiments/conv_ou16_round3_register_fed.hip (round 3: operands through registers, 12.5 us at 256 streams in
// its own clocks), experiments/conv_ou16_round6_four_waves.hip (round 6: this schedule at one wave per SIMD on 32-column tiles, 10.2 us; 8.9 here).
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

#ifndef ADK_OU16_DBG
#define ADK_OU16_DBG 0      // tuning builds only: 1 = per-workgroup wall-clock stamps (s_memrealtime, 100 MHz) of wave 0
#endif

namespace adk {

#if ADK_OU16_DBG & 1
// [workgroup 0..1023][stamp]: 0 entry, 1 the 19 LDS-DMA instructions of the wave issued, 2 first barrier passed ({biases, W1, column block 0} landed),
// 3 GEMM 1 done (72 MFMAs of 16 x 16 x 32; column blocks 1-5 streamed in beneath it), 4 act(c) written to LDS + second barrier passed, 5 GEMM 2 MFMAs and stores issued (exit)
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

typedef _Float16 f16x8u __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4u __attribute__((ext_vector_type(4)));
typedef const void __attribute__((address_space(1)))* gptr_t;
typedef void __attribute__((address_space(3)))* lptr_t;

constexpr float kOuLoScale = 2048.f, kOuLoInv = 1.f / 2048.f;
constexpr int OU_CM = 64;                        // channels between the two convs
constexpr int OU_KS2 = 2 * OU_CM / 16;           // 2 taps x 64 channels = 8 chunks of 16
constexpr int OU_TMAX = 128;                     // steps of the 1x1 conv per workgroup (8 waves x 16)

struct OuArgs { int ks1p; float inv_cout_real; int* err; unsigned out_bytes; };
typedef unsigned u32x4o __attribute__((ext_vector_type(4)));

template <int ACT>
__device__ __forceinline__ float ou_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

// Every byte of the launch arrives by LDS-DMA (global_load_lds_dwordx4: lane i's 16 bytes land at M0 + 16 i, lane-linear; the SOURCE address
// is per lane, which is where the swizzles below come from).  What the timeline of the register-fed kernel of round 3 said
// (profiles/r3_ou16_timeline.md, 256 streams): 4.4 of its 12.5 us were the ISSUE of its 48 vector-memory instructions per wave -- every lane
// loaded the B fragments of ITS time step, one wave instruction touching 32 ring rows for 32 bytes of each --, another 1.9 us the wait for ONE
// 256-byte store behind the s_waitcnt vmcnt(0) that W2's arrival needs.  With DMA the activations come in fully coalesced (whole 128-byte
// lines), no lane holds staged activations, and no ordinary vector load is left in the kernel -- biases and the history row of c arrive
// through one more DMA instruction -- so every s_waitcnt vmcnt is written here, counted, and none drains W2 early.
#define OU_DMA16(gptr, m0val) do { unsigned m0_keep_; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                                                            : "=&s"(m0_keep_) : "v"(gptr), "s"(m0val) : "memory"); } while (0)      /* M0 is the compiler's: put back */
constexpr int OU_NCB = 6;                        // column blocks of the 1x1 conv's input: 192 channels / 32

// ================================================================================================
// conv_ou16_w8_kernel: eight waves, 16-column tiles on v_mfma_f32_16x16x32_f16 (round 6, second form).
//
// The four-wave kernel of this round (experiments/conv_ou16_round6_four_waves.hip) spent 7 of its 10 us in three dependent phases -- GEMM 1,
// act / split of c, GEMM 2 -- at ONE wave per SIMD: 2.2 us of matrix-core work, the rest is every wait of a single in-order instruction
// stream (profiles/r6_ou16_timeline.md).  Here a workgroup has eight waves, two per SIMD, each with a time tile of 16 steps: the same MFMA
// cycles per SIMD (16 x 16 x 32 has half the flops of 32 x 32 x 16 in half the cycles), half the dependent chain per wave, and a second
// wave to issue while the first one waits.  The weights stay in the 32 x 32 x 16 fragment order the host packs (one layout for every
// kernel): a lane of the 16-row A operand reads ITS 16 bytes from the fragment that holds them (rows 16 (m & 1) + lane % 16, k-half
// lane / 16 & 1 of 16-k chunk 2 q + lane / 32: the 16 lanes of a ds_read_b128 group still hit 16 different bank slots).  Everything else
// as in the four-wave form: every byte by LDS-DMA, the activations through a wave-private swizzled ring of three column blocks (16 rows x
// 128 B = two DMA instructions each), hand-counted vmcnt waits, the history row of c stored last, one accumulator set per m-tile in GEMM 2
// and a branch-free finish.
// A 16 x 16 x 32 MFMA sums 32 products per instruction where the 32 x 32 x 16 form sums 16: per output element the same products and the
// same chunk order, another grouping of the f32 additions -- the result agrees with the four-wave kernel and with conv_sk16 + conv_up16 to
// f32 round-off (<= 1e-6 on the O(1) outputs), not bit for bit; the same call is bit-reproducible.
typedef float f32x4m __attribute__((ext_vector_type(4)));
constexpr int OU8_SLOT = 2048;                   // one column block of a wave's tile: 16 rows x 128 B
constexpr int OU8_RING = 3 * OU8_SLOT;           // ring of a wave; after GEMM 1 its first 16 * OU8_RSC bytes hold the wave's rows of act(c)
constexpr int OU8_RSC = 4 * OU_CM + 32;          // row stride of act(c): [64 halfs hi][64 halfs lo][32 B pad] = 288 B = 18 x 16 B: rows 2 slots apart -> conflict-free reads

template <int N> __device__ __forceinline__ void ou8_wait_vm() {
    static_assert(N >= 0 && N <= 63, "vmcnt is six bits");
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
    else if constexpr (N == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else if constexpr (N == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
    else if constexpr (N == 6) asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
    else if constexpr (N == 7) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
    else if constexpr (N == 8) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if constexpr (N == 9) asm volatile("s_waitcnt vmcnt(9)" ::: "memory");
    else if constexpr (N == 10) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
    else static_assert(N == 0, "add the count");
}

// swizzle of the 16-byte pieces of ring row n (0..15): the two k-groups a ds_read_b128 lane group mixes sit in complementary row sets
// {0-3, 12-15} / {4-11} and two pieces apart; f maps the second set's pieces onto the other half of the bank row
__device__ __forceinline__ unsigned ou8_swz(int n) { const unsigned u = (unsigned)(n >> 1) & 7u; return u ^ ((u ^ (u >> 1)) & 2u); }

template <int ACT, int MT2>
__global__ __launch_bounds__(512, 2) void conv_ou16_w8_kernel(ConvArgs a1, ConvArgs a2, OuArgs u) {
    constexpr int KS1 = 2 * OU_NCB;                         // 16-k chunks of W1 per 32-row m-tile
    constexpr int M2 = 2 * MT2;                             // 16-row m-tiles of GEMM 2
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l15 = lane & 15, kg = lane >> 4;
    const int b = blockIdx.x;
    const int T = a1.t_out;
    const int t = wave * 16 + l15;
    const bool valid = t < T;
    OU_STAMP(0);

    constexpr int W1B = 2 * KS1 * 2048, W2B = MT2 * OU_KS2 * 2048;
    unsigned char* w1l = lds;
    unsigned char* w2l = lds + W1B;
    unsigned char* ring = w2l + W2B + wave * OU8_RING;     // this wave's ring; later its rows of act(c): row r at r * OU8_RSC
    unsigned char* halo = w2l + W2B + 8 * OU8_RING;        // [8][OU8_RSC]: act(c) of the step in front of wave w's first one (w = 0: the history row)
    unsigned char* stage0 = halo + 8 * OU8_RSC;            // 1 KiB: {c[-1] | bias 2 | bias 1}, fetched by wave 0
    float* clast = reinterpret_cast<float*>(stage0 + 1024);
    typedef unsigned char __attribute__((address_space(3)))* lds_u8_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_u8_t)lds;
    const unsigned lane16 = (unsigned)lane * 16u;
    const bool act = __builtin_amdgcn_readfirstlane(wave * 16 < T ? 1 : 0) != 0;     // a wave whose 16 steps all lie past the end only helps with the weights

    // ---- every byte by LDS-DMA, oldest first: {c[-1], biases} (wave 0 only), W1 (6 per wave), column blocks 0-2 (2 each); blocks 3-5 follow from
    // inside GEMM 1 as ring slots fall free, W2 (2 MT2 per wave) behind block 5's request -- it is needed last.  The hand-counted waits say how
    // many of THIS wave's instructions may still be in flight (wave 0's extra one is its oldest: the same counts hold for it).
    // (W1 cut into the 8 KiB each column block multiplies and requested block by block with a barrier per block, so that GEMM 1 could start on
    // 24 KiB: measured slower, 9.2 against 8.7-9.0 us -- the issue of DMA instructions itself proceeds at the CU's fill rate, ~24 ns per KiB
    // with 256 workgroups at it, so what is asked for first is there first either way and the extra barriers only add waits.  Also measured, no
    // gain: block 0 -- the cold bytes; the weights sit in the L2s -- requested in front of W1 (9.0); W1 in two halves of K, the second behind
    // blocks 0-2, one more barrier (9.2).  experiments/sessions/r6_s17.sh, runs 18-21.)
    if (wave == 0) {
        const unsigned char* src = reinterpret_cast<const unsigned char*>(a1.wfrag) + lane16;
        if (lane < 16) src = reinterpret_cast<const unsigned char*>(a2.in + ((size_t)b * a2.in_rows + a2.in_row0) * a2.in_ch + a2.in_choff) + lane16;
        else if (lane < 16 + 8 * MT2) { if (a2.bias) src = reinterpret_cast<const unsigned char*>(a2.bias) + (lane - 16) * 16; }
        else if (lane >= 40 && lane < 56) { if (a1.bias) src = reinterpret_cast<const unsigned char*>(a1.bias) + (lane - 40) * 16; }
        OU_DMA16(src, lds0 + (unsigned)(stage0 - lds));
    }
    {
        const unsigned char* g1 = reinterpret_cast<const unsigned char*>(a1.wfrag) + (size_t)tid * 16;
        const unsigned l1 = lds0 + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < W1B / 8192; ++i) OU_DMA16(g1 + 8192 * i, l1 + 8192u * i);
    }
    const unsigned char* xsrc[2];
#pragma unroll
    for (int j = 0; j < 2; ++j) {
        const int r = 8 * j + (lane >> 3);
        int tr = wave * 16 + r;
        if (tr >= T) tr = T - 1;                            // rows past the end: a copy of the last one; nothing of theirs is stored
        int row = a1.in_row0 + tr;
        if (row >= a1.in_rows) row -= a1.in_rows;
        xsrc[j] = reinterpret_cast<const unsigned char*>(a1.in + ((size_t)b * a1.in_rows + row) * a1.in_ch + a1.in_choff) + 16 * ((unsigned)(lane & 7) ^ ou8_swz(r));
    }
    const unsigned ring0 = lds0 + (unsigned)(ring - lds);
    auto issue_block = [&](int cb) __attribute__((always_inline)) {
        const unsigned dst = ring0 + (unsigned)(cb % 3) * OU8_SLOT;
#pragma unroll
        for (int j = 0; j < 2; ++j) OU_DMA16(xsrc[j] + 128 * cb, dst + 1024u * j);
    };
    constexpr int W2N = W2B / 8192;                        // W2 instructions per wave: 2 * MT2
    auto issue_w2 = [&]() __attribute__((always_inline)) {
        const unsigned char* g2 = reinterpret_cast<const unsigned char*>(a2.wfrag) + (size_t)tid * 16;
        const unsigned l2 = lds0 + (unsigned)W1B + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < W2N; ++i) OU_DMA16(g2 + 8192 * i, l2 + 8192u * i);
    };
    const float* b2l = reinterpret_cast<const float*>(stage0 + 256);
    const float* b1l = reinterpret_cast<const float*>(stage0 + 640);
    // this lane's A-fragment address inside a packed 32-row m-tile: rows 16 * (m16 & 1) + l15, k-half kg & 1 of 16-k chunk 2 q + (kg >> 1)
    const unsigned a_lane = (unsigned)(l15 + 32 * (kg & 1)) * 16u + (unsigned)(kg >> 1) * 2048u;
    float chk = 0.f;                                        // stays 0 while every output is finite

    if (!act) {
        OU_STAMP(1);
        ou8_wait_vm<0>(); __syncthreads();                 // its pieces of W1 have landed
        OU_STAMP(2);
        issue_w2();
        ou8_wait_vm<0>();
        OU_STAMP(3);
    } else {
        // ---- GEMM 1: c[m][t] = sum_k W1[m][k] x[k][t], 64 rows (four 16-row m-tiles) x this wave's 16 steps; one column block = one 32-k step ----
        issue_block(0); issue_block(1); issue_block(2);
        OU_STAMP(1);
        f32x4m am[4], ac[4];
#pragma unroll
        for (int m = 0; m < 4; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) { am[m][e] = 0.f; ac[m][e] = 0.f; }
        const unsigned swz = ou8_swz(l15);
        // (A fragments requested one step ahead in two register sets, before the wait for the next block: measured slower, 8.9 -> 9.4 us --
        // as in the four-wave form the compiler's own placement of the fragment reads is the better one)
#pragma unroll
        for (int cb = 0; cb < OU_NCB; ++cb) {
            // in flight at most (12 issued up front, + 2 per block requested in the loop, + W2N behind block 5):
            if (cb == 0) { ou8_wait_vm<4>(); __syncthreads(); }                 // {W1, block 0} landed here -- and, behind the barrier, everybody's W1
            else if (cb == 1 || cb == 2) ou8_wait_vm<4>();
            else if (cb == 3) ou8_wait_vm<4 + W2N>();                           // blocks 4, 5 and W2 may stay in flight
            else if (cb == 4) ou8_wait_vm<2 + W2N>();
            else ou8_wait_vm<W2N>();
            if (cb == 0) {
                OU_STAMP(2);
                if (tid < OU_CM / 4) {                     // history row: activation, split, into the halo row of wave 0
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
            }
            const unsigned char* xs = ring + (cb % 3) * OU8_SLOT + l15 * 128;
            const float4 x0 = *reinterpret_cast<const float4*>(xs + 16 * ((unsigned)(2 * kg) ^ swz));
            const float4 x1 = *reinterpret_cast<const float4*>(xs + 16 * ((unsigned)(2 * kg + 1) ^ swz));
            if (cb + 3 < OU_NCB) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads of this slot have returned: it may be overwritten
                issue_block(cb + 3);
                if (cb + 3 == OU_NCB - 1) issue_w2();
            }
            const float x[8] = {x0.x, x0.y, x0.z, x0.w, x1.x, x1.y, x1.z, x1.w};
            f16x8u bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)x[e];
                bh[e] = h; bl[e] = (_Float16)((x[e] - (float)h) * kOuLoScale);
            }
            f16x8u Ah[4], Al[4];
#pragma unroll
            for (int m = 0; m < 4; ++m) {
                const unsigned char* wp = w1l + (size_t)((m >> 1) * KS1 + 2 * cb) * 2048 + (m & 1) * 256 + a_lane;
                Ah[m] = *reinterpret_cast<const f16x8u*>(wp);
                Al[m] = *reinterpret_cast<const f16x8u*>(wp + 1024);
            }
#pragma unroll
            for (int m = 0; m < 4; ++m) am[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[m], bh, am[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) ac[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah[m], bl, ac[m], 0, 0, 0);
#pragma unroll
            for (int m = 0; m < 4; ++m) ac[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al[m], bh, ac[m], 0, 0, 0);
        }
        ou8_wait_vm<0>();                                  // W2
        OU_STAMP(3);
        // c (+ bias): act(c), split, over this wave's ring; the last step's raw row to `clast`; the row the NEXT wave's first step needs as
        // its older tap to that wave's halo row.  Lane (column l15, kg) holds channels 16 m + 4 kg + {0..3}.
        unsigned char* lrow = ring + l15 * OU8_RSC;
        unsigned char* hrow2 = halo + (wave + 1) * OU8_RSC;
        const bool is_last = valid && t == T - 1;
        const bool to_halo = l15 == 15 && wave < 7;
#pragma unroll
        for (int m = 0; m < 4; ++m) {
            const int ml = 16 * m + 4 * kg;
            float v[4] = {fmaf(ac[m][0], kOuLoInv, am[m][0]), fmaf(ac[m][1], kOuLoInv, am[m][1]), fmaf(ac[m][2], kOuLoInv, am[m][2]), fmaf(ac[m][3], kOuLoInv, am[m][3])};
            chk = fmaf(v[0], 0.f, chk); chk = fmaf(v[1], 0.f, chk); chk = fmaf(v[2], 0.f, chk); chk = fmaf(v[3], 0.f, chk);
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
            *reinterpret_cast<f16x4u*>(lrow + 2 * ml) = hi;
            *reinterpret_cast<f16x4u*>(lrow + 2 * OU_CM + 2 * ml) = lo;
            if (to_halo) {
                *reinterpret_cast<f16x4u*>(hrow2 + 2 * ml) = hi;
                *reinterpret_cast<f16x4u*>(hrow2 + 2 * OU_CM + 2 * ml) = lo;
            }
        }
    }
    __syncthreads();                                       // (W2 landed with the last column block: vmcnt(0) above)
    OU_STAMP(4);

    if (act) {
        // ---- GEMM 2: the polyphase transposed conv; k = (tap j, channel): 32-k step q = tap q / 2, channels 32 (q & 1) ..; tap 0 = c[t-1], tap 1 = c[t] ----
        int orow0 = a2.out_cursor + t * a2.up;
        orow0 %= a2.out_rows;
        const unsigned char* xr0 = (l15 == 0 ? halo + wave * OU8_RSC : ring + (l15 - 1) * OU8_RSC) + 16 * kg;
        const unsigned char* xr1 = ring + l15 * OU8_RSC + 16 * kg;
        f16x8u bh[4], bl[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const unsigned char* p = (q >> 1 ? xr1 : xr0) + 64 * (q & 1);
            bh[q] = *reinterpret_cast<const f16x8u*>(p);
            bl[q] = *reinterpret_cast<const f16x8u*>(p + 2 * OU_CM);
        }
        const __amdgpu_buffer_rsrc_t rsrc_out = __builtin_amdgcn_make_buffer_rsrc(a2.out, 0, u.out_bytes, 0x00020000);
        const unsigned out_base = ((unsigned)b * (unsigned)a2.out_rows * (unsigned)a2.out_ch + (unsigned)a2.out_choff) * 4u;
        const unsigned row_bytes = (unsigned)a2.out_ch * 4u;
        const bool has_b2 = a2.bias != nullptr;
        const unsigned oob_mask = valid ? 0u : 0x80000000u;    // columns past the end: the store goes out of bounds (dropped), no branch
        f32x4m am2[M2], ac2[M2];
#pragma unroll
        for (int m = 0; m < M2; ++m)
#pragma unroll
            for (int e = 0; e < 4; ++e) { am2[m][e] = 0.f; ac2[m][e] = 0.f; }
#pragma unroll
        for (int m = 0; m <= M2; ++m) {
            if (m < M2) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const unsigned char* wp = w2l + (size_t)((m >> 1) * OU_KS2 + 2 * q) * 2048 + (m & 1) * 256 + a_lane;
                    const f16x8u Ah = *reinterpret_cast<const f16x8u*>(wp);
                    const f16x8u Al = *reinterpret_cast<const f16x8u*>(wp + 1024);
                    am2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, bh[q], am2[m], 0, 0, 0);
                    ac2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Ah, bl[q], ac2[m], 0, 0, 0);
                    ac2[m] = __builtin_amdgcn_mfma_f32_16x16x32_f16(Al, bh[q], ac2[m], 0, 0, 0);
                }
            }
            if (m > 0) {
                const int m1 = m - 1;                          // finish of the previous m-tile: GEMM rows 16 m1 + 4 kg + {0..3}
                const int ml = 16 * m1 + 4 * kg;
                float4 v = make_float4(fmaf(ac2[m1][0], kOuLoInv, am2[m1][0]), fmaf(ac2[m1][1], kOuLoInv, am2[m1][1]),
                                       fmaf(ac2[m1][2], kOuLoInv, am2[m1][2]), fmaf(ac2[m1][3], kOuLoInv, am2[m1][3]));
                chk = fmaf(v.x, 0.f, chk); chk = fmaf(v.y, 0.f, chk); chk = fmaf(v.z, 0.f, chk); chk = fmaf(v.w, 0.f, chk);
                float4 bb = *reinterpret_cast<const float4*>(b2l + ml);
                bb.x = has_b2 ? bb.x : 0.f; bb.y = has_b2 ? bb.y : 0.f; bb.z = has_b2 ? bb.z : 0.f; bb.w = has_b2 ? bb.w : 0.f;
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
                const int ph = (int)(((float)ml + 0.5f) * u.inv_cout_real);        // ml / cout_real, exact for these sizes
                int r2 = orow0 + ph;
                if (r2 >= a2.out_rows) r2 -= a2.out_rows;
                const unsigned off = (out_base + (unsigned)r2 * row_bytes + (unsigned)(ml - ph * a2.cout_real) * 4u) | oob_mask;
                u32x4o pv;
                pv.x = __float_as_uint(v.x); pv.y = __float_as_uint(v.y); pv.z = __float_as_uint(v.z); pv.w = __float_as_uint(v.w);
                // sc1 = write-through: the 64-byte pieces (four lanes per step and m-tile) leave the L2 while the launch runs instead of in its
                // end-of-kernel write-back -- 11.3 -> 11.0 us by rocprofv3, nothing changes in the three-stream schedule (sessions r6_s27.sh,
                // r6_s29.sh); nt is slower, and the same policy on conv_up16's 32-byte pieces costs 27 -> 42 us (r6_s28.sh)
                __builtin_amdgcn_raw_buffer_store_b128(pv, rsrc_out, off, 0, 16 /* sc1 */);
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

bool ou_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// a1: the 1x1 conv into 64 channels, a2: the 2-tap polyphase transposed conv that reads exactly what a1 writes
bool conv_ou16_fusable(const ConvArgs& a1, const ConvArgs& a2) {
    if (!conv_up16_supported(a2) || a2.act_in == ADK_ACT_TANH) return false;
    if (!a1.wfrag || a1.taps != 1 || a1.stride != 1 || a1.up != 1 || a1.groups != 1 || a1.res || a1.act_in != ADK_ACT_NONE || a1.act_out != ADK_ACT_NONE) return false;
    if (a1.cout_g != OU_CM || a1.cin_g != 32 * OU_NCB) return false;                   // instantiated: 192 -> 64 (six column blocks of 32 channels)
    if ((unsigned long long)a2.batch * a2.out_rows * a2.out_ch * 4ull >= 0x7fffffffull) return false;      // buffer stores: 32-bit byte offsets
    if (a1.batch != a2.batch || a1.t_out != a2.t_out || a1.t_out < 1 || a1.t_out > OU_TMAX) return false;
    if (a2.in != a1.out || a2.in_rows != a1.out_rows || a2.in_ch != a1.out_ch || a2.in_choff != a1.out_choff) return false;
    if (a2.in_row0 != (a1.out_cursor + a1.out_rows - 1) % a1.out_rows) return false;    // one row of history, right in front of the new rows
    if ((a1.in_ch % 4) || (a1.in_choff % 4) || (a1.out_ch % 4) || (a1.out_choff % 4) || !ou_aligned16(a1.in) || !ou_aligned16(a1.out) || !ou_aligned16(a1.wfrag)) return false;
    if (a1.bias && !ou_aligned16(a1.bias)) return false;
    return true;
}

int launch_conv_ou16(const ConvArgs& a1, const ConvArgs& a2, hipStream_t s) {
    if (!conv_ou16_fusable(a1, a2)) return ADK_ERR_STATE;
    if (a1.n_total == 0) return ADK_OK;
    OuArgs u;
    u.ks1p = (a1.ktot + 63) / 64 * 4;
    u.inv_cout_real = 1.0f / (float)a2.cout_real;
    u.err = conv_err_word(a1);
    u.out_bytes = (unsigned)((unsigned long long)a2.batch * a2.out_rows * a2.out_ch * 4ull);        // < 2^31: conv_ou16_fusable
    const int mt2 = a2.cout_g / 32;
    const size_t lds = (size_t)2 * u.ks1p * 2048 + (size_t)mt2 * OU_KS2 * 2048 + (size_t)8 * OU8_RING + (size_t)8 * OU8_RSC + 1024 + OU_CM * sizeof(float);
    // a function attribute belongs to ONE instantiation (and one device): the flags are keyed by the (ACT, MT2) pair -- every
    // instantiation decays to the same pointer type, so a flag inside a generic lambda over that pointer would be shared by all of them
    auto go = [&](auto act, auto mt) -> int {
        constexpr int ACT = decltype(act)::value, MT2 = decltype(mt)::value;
        auto kern = conv_ou16_w8_kernel<ACT, MT2>;
        static bool attr_set_dev[kMaxDevices] = {};         // one array per (ACT, MT2): the lambda's operator() is instantiated per tag type pair
        bool& attr_set = attr_set_dev[current_device()];
        if (!attr_set) {
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
            attr_set = true;
        }
        hipLaunchKernelGGL(kern, dim3((unsigned)a1.batch), dim3(512), lds, s, a1, a2, u);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    auto by_mt = [&](auto act) -> int {
        if (mt2 == 1) return go(act, std::integral_constant<int, 1>());
        if (mt2 == 2) return go(act, std::integral_constant<int, 2>());
        return go(act, std::integral_constant<int, 3>());
    };
    if (a2.act_in == ADK_ACT_ELU) return by_mt(std::integral_constant<int, ADK_ACT_ELU>());
    if (a2.act_in == ADK_ACT_LEAKY) return by_mt(std::integral_constant<int, ADK_ACT_LEAKY>());
    return by_mt(std::integral_constant<int, ADK_ACT_NONE>());
}

}  // namespace adk
