// Implicit-GEMM causal conv on the fp32 matrix cores (v_mfma_f32_32x32x2_f32: exact f32 FMA chain),
// persistent "stream-K" schedule.
//
//   D[m][n] = sum_k W[m][k] * act(X[k][n]),   m = output channel (x phase for transposed convs),
//   n = (stream b, output step t),            k = (tap j, input channel ci)
//
// Replaces F.conv1d / F.conv_transpose1d as called from CausalConv1d.inference /
// CausalConvTranspose1d.inference (layers/conv_layer.py:153-156, 194-197) and Conv1d1x1 (:28-32),
// fused with the reference's surrounding element-wise ops (input ELU/LeakyReLU, bias, residual add).
//
// Many independent streams make N = B*T large even when one stream contributes a single step, so
// every conv of the path with Cin % 32 == 0 is a GEMM with M in 32..1280, K in 32..2816 -- but the
// tile counts (60..2700) do not divide the 256 CUs, and the deep layers have few tiles with very
// long K.  Hence:
//   * grid = G persistent workgroups (G = 256 CUs x occupancy); the (tile, K-chunk) work units are
//     split EVENLY over them in tile order (stream-K).  A workgroup that covers a whole tile writes
//     it out directly; a tile cut by range boundaries is finished by the workgroup that holds its first
//     chunk (the owner): the later ranges publish their raw partial accumulators FIRST THING in their run
//     (write-through stores + one flag), the owner adds them in range order (deterministic) at the END of
//     its run and runs the epilogue.  Progress: workgroups are dispatched in blockIdx order and an owner
//     only ever waits for higher ranges, which publish before doing anything else -- so whenever a slot
//     frees, the next workgroup to start is exactly one some spinner may be waiting for; the spin is
//     bounded anyway (flag bit 1) so a broken assumption cannot hang the device.  (A wait-free variant --
//     every contributor publishes, the last arriver reduces -- was measured 3.4 us per launch slower: the
//     owner's partial then also takes the write-through round trip on the critical path.)
//   * ranges are XCD-contiguous (block b -> range (b%8)*(G/8)+b/8) and tiles are ordered with the
//     M-tile fastest, so the workgroups sharing one XCD's L2 walk neighbouring tiles: the weight
//     panel of a group stays L2-resident, the activations stream from HBM once.
//   * W is pre-packed in MFMA-fragment order (adk_pack_weights_mfma) and goes global -> VGPR
//     directly, one coalesced 1 KiB wave-load per 32x8 fragment, prefetched a chunk ahead; it never
//     touches LDS.
//   * X columns are gathered from the channel-last state rings, 128 contiguous bytes per (column,
//     tap): coalesced along the channel axis, the causal history is just earlier rows.  The input
//     activation is applied once per staged element; the chunk lands in LDS (row stride 36 floats:
//     conflict-free ds_write_b128 / ds_read_b128), double buffered, one barrier per 32-deep chunk,
//     next chunk's global loads in flight under the MFMAs -- also across tile boundaries.
//   * each lane reads 4 consecutive k per ds_read_b128 and feeds 4 MFMAs per accumulator (the k
//     order inside a chunk is permuted identically for both operands: only the order of the exact
//     f32 sum changes).
//
// SPLIT = true is the opt-in split-precision variant (ADK_IMPL_SPLIT16*): same schedule, same fix-up, but
// the staged X chunk is stored as f16 hi / f16 lo*2048 halves (same LDS bytes), the weights come in the
// adk_pack_weights_split16 layout (same 8 KiB per (32 rows, 64 k)), and a 64-deep chunk is 4 x 3
// v_mfma_f32_32x32x16_f16 per accumulator pair (hi*hi -> main, hi*lo + lo*hi -> cross) instead of
// 32 v_mfma_f32_32x32x2_f32; main + cross/2048 is formed when a segment ends.  See conv_rl16.hip for the
// error analysis.
#include "adk_common.h"
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <atomic>
#include <mutex>

#ifndef ADK_SK_SC1_READ
#define ADK_SK_SC1_READ 1   // 1: partial tiles read with sc1 loads, no acquire fence; 0: plain loads behind an agent-scope acquire
#endif
#ifndef ADK_SK_OWNER_LATE
#define ADK_SK_OWNER_LATE 1 // 1: with two workgroups per tile, the later-dispatched blocks of an XCD take the owner halves
#endif
#ifndef ADK_SK16_DBG
#define ADK_SK16_DBG 0      // tuning experiments only: 1 = lo part not computed, 2 = no input activation, 64 = half of the waves load no weights (knock-out)
#endif

namespace adk {

#if ADK_SK16_DBG & 32
// per-workgroup wall-clock stamps (s_memrealtime, 100 MHz): kernel entry, loop entry, loop exit, kernel exit of wave 0
__device__ unsigned long long g_sk_wg_trace[512 * 4];
#endif
#if ADK_SK16_DBG & 16
// timeline of one workgroup's wave 0 (s_memtime at fixed points of every iteration): tools/kbench prints it (adk_debug_sk_trace)
__device__ unsigned long long g_sk_trace[64 * 8];
#define SK_STAMP(slot) do { if (trace_on && it < 64) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                            __builtin_amdgcn_sched_barrier(0); if (lane == 0) trace_lds[it * 8 + (slot)] = t_; } } while (0)
#else
#define SK_STAMP(slot) do { } while (0)
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8s __attribute__((ext_vector_type(8)));
typedef _Float16 f16x4s __attribute__((ext_vector_type(4)));
constexpr float kSkLoScale = 2048.f, kSkLoInv = 1.f / 2048.f;

constexpr int KC = 64;    // K chunk: two 32-channel half-chunks (each one tap x 32 channels)
constexpr int kGvCounters = 4096;      // arrival counters of conv_gv16 at the end of the workspace: one per (group, 32-row m-tile) of a launch

struct SkArgs {
    float* ws;            // partial-tile workspace: [G][256 threads][NJ*16] floats
    unsigned* flags;      // [G] publish flags: flags[r] == epoch <=> range r's head partial is in ws
    unsigned epoch;       // unique per launch on this workspace (never 0)
    int G;                // persistent workgroups (ranges); the grid is 8 * ceil(G / 8) blocks
    int split;            // > 0: tile-aligned ranges, `split` workgroups per tile (range r = tile r / split, part r % split)
    int owner_chunks;     // split == 2: chunks of a tile that its owner (first half) takes
    int tpw;              // > 0: tile-aligned ranges, `tpw` whole tiles per workgroup;  both 0: total / G units each, wherever that cuts
    int tiles;
    int m_tiles, n_tiles, nchunks;
    int cpt;              // 32-channel blocks per tap = cin_g / 32
    int kgroups;          // 8-k fragments per 32-row m-tile, K zero-padded to a multiple of 64
    int mt32_per_g;       // 32-row fragment tiles per group
    unsigned in_bytes, w_bytes;   // buffer-descriptor extents of the input arena view / packed weights
    unsigned ws_bytes;            // extent of the partial workspace
    int* err;                     // device error word (bit 1: a publish flag never arrived; bit 3: split-f16 operand overflow)
    float inv_t_out;
    long long total;      // tiles * nchunks
    int epi_lds;          // 1: finished tiles go through LDS to memory (sk_epilogue_lds), 0: straight from the accumulators (sk_epilogue)
};

constexpr int kActPre = 100;   // ACT template value of the stream-K kernel: the staged pieces come from a shadow ring, already activated and split

template <int ACT>
__device__ __forceinline__ float act_in_apply(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

// first work unit of range r (r = G: one past the last) -- shared by the kernels and adk_streamk_range_start
__host__ __device__ __forceinline__ long long sk_range_start(int r, int G, int split, int tpw, int tiles, int nchunks, int owner_chunks, long long total) {
    if (split > 0) {
        const int t = r / split, p = r - t * split;
        if (split == 2) return (long long)t * nchunks + (p ? owner_chunks : 0);
        return (long long)t * nchunks + (p * nchunks) / split;
    }
    if (tpw > 0) {
        long long t = (long long)r * tpw;
        if (t > tiles) t = tiles;
        return t * nchunks;
    }
    return (long long)r * total / G;
}
__device__ __forceinline__ long long sk_u0(int r, const SkArgs& sk) {
    return sk_range_start(r, sk.G, sk.split, sk.tpw, sk.tiles, sk.nchunks, sk.owner_chunks, sk.total);
}

// Epilogue for one wave's 32 x (32*NJ) accumulator block: bias, residual, output activation, store.
// Lane holds column n = n0w + 32*j + (lane&31) and rows ml0 + 8*qd + 4*(lane>>5) + {0..3}.
// CHECK (split-f16 variant): an operand beyond the f16 range became inf when it was split, so every output it feeds is
// non-finite -- testing the outputs once here replaces a compare per staged element per tap (5 % of the kernel time).
template <int NJ, bool CHECK = false>
__device__ __forceinline__ void sk_epilogue(const ConvArgs& a, const f32x16 (&acc)[NJ], int g, int ml0, int n0w, int lane, int* err = nullptr) {
    const int l31 = lane & 31, lh = lane >> 5;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
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
            if (CHECK) bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
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
            if (CHECK && a.out_sh) {
                // the readers' operand form of these 4 channels, once: act, split into f16 hi / lo*2048 (what lstore_piece computes per
                // staged element otherwise).  Shadow row layout: per 8 channels 32 bytes, [8 x f16 hi][8 x f16 lo] -- an MFMA B fragment
                // (8 consecutive k of one column) is then ONE aligned 16-byte piece, for the stream-K kernel's staging as for the
                // DMA-fed kernel below; this lane's 4 channels are half of such a group: two 8-byte stores
                const float x[4] = {act_apply(v.x, a.sh_act, a.sh_slope), act_apply(v.y, a.sh_act, a.sh_slope),
                                    act_apply(v.z, a.sh_act, a.sh_slope), act_apply(v.w, a.sh_act, a.sh_slope)};
                f16x4s hi, lo;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 h = (_Float16)x[e];
                    hi[e] = h;
                    lo[e] = (_Float16)((x[e] - (float)h) * kSkLoScale);
                }
                unsigned char* srow = reinterpret_cast<unsigned char*>(a.out_sh + ((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff);
                unsigned char* sp = srow + (size_t)(ocol >> 3) * 32 + (ocol & 4) * 2;           // group ocol / 8, first or second half of its hi block
                *reinterpret_cast<f16x4s*>(sp) = hi;
                *reinterpret_cast<f16x4s*>(sp + 16) = lo;
            }
        }
    }
    if (CHECK && bad) atomicOr(err, 8);
}

// tile id -> (group, m-tile, n-tile); M-tile fastest so neighbouring tiles share the X columns and
// all tiles of a group share the (L2-resident) weight panel
__device__ __forceinline__ void sk_tile_coords(int tile, const SkArgs& sk, int& g, int& mt, int& nt) {
    mt = tile % sk.m_tiles;
    const int rest = tile / sk.m_tiles;
    nt = rest % sk.n_tiles;
    g = rest / sk.n_tiles;
}

typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float4 buf_load4(__amdgpu_buffer_rsrc_t rsrc, unsigned voff, unsigned soff) {
    const u32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rsrc, voff, soff, 0);
    return make_float4(__uint_as_float(v.x), __uint_as_float(v.y), __uint_as_float(v.z), __uint_as_float(v.w));
}

// n / d for 0 <= n < 2^24 via the f32 reciprocal (exact after one correction step)
__device__ __forceinline__ int fast_div(int n, int d, float inv_d) {
    int q = (int)(__int2float_rn(n) * inv_d);
    int r = n - q * d;
    if (r < 0) { --q; r += d; }
    if (r >= d) { ++q; }
    return q;
}

// The same epilogue with the tile taken through LDS first (round 4).  In the accumulator layout a lane's neighbours are 32 columns = 32
// ring rows apart: sk_epilogue writes a 32 x 32 block as 32-byte fragments, one cache line touched per lane pair and store.  Here the
// workgroup's BM x BN tile is written to T[column][BM channels] (row stride BM + 4 floats: conflict-free float4 writes) and read back
// with a thread = 8 consecutive channels of one column: every global access (residual, output, shadow) is 16 bytes per lane along the
// channel axis, a column's BM channels are whole cache lines.  Per element the operations and their order are sk_epilogue's -- the
// results are bit-identical.  (Where it came from: the tail of conv_gk16, round 4's DMA-fed 128 x 128-tile kernel, now experiments/conv_gk16.hip: 19.8 -> 4.4 us, profiles/r4_gk16_timeline.md.)
// T must hold BN * (BM + 4) floats; the caller brackets the call with the barriers that make the buffer free / keep it until read.
template <int NJ, int BM, int BN, int NT, bool CHECK>
__device__ __forceinline__ void sk_epilogue_lds(const ConvArgs& a, const f32x16 (&acc)[NJ], float* T, int g, int m0, int n0, int wm, int wn,
                                                int tid, float inv_t_out, int* err) {
    constexpr int TSF = BM + 4;
    const int lane = tid & 63, l31 = lane & 31, lh = lane >> 5;
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        float* trow = T + (size_t)(wn * NJ * 32 + j * 32 + l31) * TSF + wm * 32 + 4 * lh;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd)
            *reinterpret_cast<float4*>(trow + 8 * qd) = make_float4(acc[j][4 * qd], acc[j][4 * qd + 1], acc[j][4 * qd + 2], acc[j][4 * qd + 3]);
    }
    __syncthreads();
    constexpr int GPC = BM / 8;                          // 8-channel groups per column
    bool bad = false;
    for (int it = tid; it < BN * GPC; it += NT) {
        const int col = it / GPC, cg = it - col * GPC;
        const int n = n0 + col;
        const int ml = m0 + 8 * cg;                      // channel within the group
        if (n >= a.n_total || ml >= a.cout_g) continue;
        float4 t0 = *reinterpret_cast<const float4*>(T + (size_t)col * TSF + 8 * cg);
        float4 t1 = *reinterpret_cast<const float4*>(T + (size_t)col * TSF + 8 * cg + 4);
        if (CHECK) bad |= !(fabsf(t0.x) <= 3.0e38f) | !(fabsf(t0.y) <= 3.0e38f) | !(fabsf(t0.z) <= 3.0e38f) | !(fabsf(t0.w) <= 3.0e38f) |
                          !(fabsf(t1.x) <= 3.0e38f) | !(fabsf(t1.y) <= 3.0e38f) | !(fabsf(t1.z) <= 3.0e38f) | !(fabsf(t1.w) <= 3.0e38f);
        const int b = fast_div(n, a.t_out, inv_t_out), t = n - b * a.t_out;
        const int mg = g * a.cout_g + ml;
        if (a.bias) {
            const float4 b0 = *reinterpret_cast<const float4*>(a.bias + mg), b1 = *reinterpret_cast<const float4*>(a.bias + mg + 4);
            t0.x += b0.x; t0.y += b0.y; t0.z += b0.z; t0.w += b0.w;
            t1.x += b1.x; t1.y += b1.y; t1.z += b1.z; t1.w += b1.w;
        }
        if (a.res) {
            int rrow = a.res_cursor + t;
            if (rrow >= a.res_rows) rrow -= a.res_rows;
            const float* resp = a.res + ((size_t)b * a.res_rows + rrow) * a.res_ch + a.res_choff + g * a.res_gstride + ml;
            const float4 r0 = *reinterpret_cast<const float4*>(resp), r1 = *reinterpret_cast<const float4*>(resp + 4);
            t0.x += r0.x; t0.y += r0.y; t0.z += r0.z; t0.w += r0.w;
            t1.x += r1.x; t1.y += r1.y; t1.z += r1.z; t1.w += r1.w;
        }
        float x[8] = {t0.x, t0.y, t0.z, t0.w, t1.x, t1.y, t1.z, t1.w};
        if (a.act_out != ADK_ACT_NONE) {
#pragma unroll
            for (int e = 0; e < 8; ++e) x[e] = act_apply(x[e], a.act_out, 0.f);
        }
        int orow = a.out_cursor + t * a.up, ocol = mg;
        if (a.up > 1) { const int ph = mg / a.cout_real; orow += ph; ocol = mg - ph * a.cout_real; }
        if (orow >= a.out_rows) orow -= a.out_rows;
        const size_t oidx = ((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff + ocol;
        *reinterpret_cast<float4*>(a.out + oidx) = make_float4(x[0], x[1], x[2], x[3]);
        *reinterpret_cast<float4*>(a.out + oidx + 4) = make_float4(x[4], x[5], x[6], x[7]);
        if (CHECK && a.out_sh) {
            f16x8s hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = act_apply(x[e], a.sh_act, a.sh_slope);
                const _Float16 h = (_Float16)y;
                hi[e] = h;
                lo[e] = (_Float16)((y - (float)h) * kSkLoScale);
            }
            unsigned char* sp = reinterpret_cast<unsigned char*>(a.out_sh + ((size_t)b * a.out_rows + orow) * a.out_ch + a.out_choff) + (size_t)(ocol >> 3) * 32;
            *reinterpret_cast<f16x8s*>(sp) = hi;
            *reinterpret_cast<f16x8s*>(sp + 16) = lo;
        }
    }
    if (CHECK && bad) atomicOr(err, 8);
}

// Main kernel.  One iteration = one 64-deep K chunk (two 32-channel half-chunks, which may belong to
// different taps): 32*NJ MFMAs per wave (>= 4096 cycles) between barriers, so the global loads issued
// at the top of an iteration (measured latency under MFMA load ~1.5 us) have landed when the bottom
// of the iteration stores them to LDS.  With one or two waves per SIMD every instruction between two
// MFMA bursts is exposed, so per-chunk addressing is reduced to buffer loads with one per-thread VGPR
// offset per staged column (updated by adds) and scalar offsets for the weight stream; columns past
// N and the zero-padded K tail read out of bounds (= 0) instead of branching.
// KD = K-chunk depth in units of 64 (1: 64-deep, 2: 128-deep).  The split-f16 variant spends so little time in the
// matrix cores per 64 k that the per-iteration costs (barrier, address update, exposed load latency) dominate: KD = 2
// puts twice the bytes in flight per iteration and halves the iteration count.
// WGM * WGN = 8 (512 threads, one workgroup per CU -- still 8 waves per CU): the 128 x 128 tile of the split variant, half the
// operand bytes per MFMA of the 64 x 64 tile at the same number of waves in flight.
template <int WGM, int WGN, int NJ, int ACT, bool SPLIT, int KD>
__global__ __launch_bounds__(64 * WGM * WGN, (WGM * WGN == 4) ? 2 : 1) void conv_sk_kernel(ConvArgs a, SkArgs sk) {
    constexpr int NT = 64 * WGM * WGN;                // threads per workgroup
    constexpr int BN = 32 * NJ * WGN;
    constexpr int KCC = KC * KD;                      // chunk depth
    constexpr int LDK = KCC + 4;                      // padded LDS row stride (floats); 68 and 132 are both = 4 mod 64
    constexpr int QPC = 16 * KD;                      // 16-byte pieces per staged column
    constexpr int CPR = NT / QPC;                     // columns staged per round
    constexpr int RB = BN / CPR;                      // staging rounds
    static_assert(WGM * WGN == 4 || WGM * WGN == 8, "4 or 8 waves");
    static_assert(BN % CPR == 0, "staging rounds");
    static_assert(SPLIT || KD == 1, "the exact-f32 variant keeps 64-deep chunks");
    extern __shared__ __attribute__((aligned(16))) float Bs[];   // [2][BN*LDK]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WGN, wn = wave % WGN;
    const int l31 = lane & 31, lh = lane >> 5;
    const int srow = tid / QPC, quad = tid % QPC;     // staged column (mod CPR) and 16-byte piece of the KCC-float row
    const int half = quad >> 3;                       // which 32-channel sub-chunk (0 .. 2*KD-1) this thread stages

    // XCD-contiguous range of work units
    const int per_xcd = (sk.G + 7) >> 3;
    int r = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
#if ADK_SK_OWNER_LATE
    // Two workgroups per tile, two per CU: the second workgroup placed on a CU runs ~20 % slower than the first (per-workgroup
    // wall clocks, profiles/r2_sk16_timeline.md), and the owner of a tile has to wait for the other half anyway.  Blocks are
    // dispatched in index order, so let the first half of an XCD's blocks take the contributor halves (odd ranges) and the second
    // half the owner halves: the partial tile is there long before its owner asks for it.
    if (sk.split == 2 && !(per_xcd & 1) && sk.G == 8 * per_xcd) {
        const int j = (int)(blockIdx.x >> 3), hp = per_xcd >> 1;
        r = (int)(blockIdx.x & 7) * per_xcd + (j < hp ? 2 * j + 1 : 2 * (j - hp));
    }
#endif
    if (r >= sk.G) return;
#if ADK_SK16_DBG & 32
    const unsigned long long wg_t0 = __builtin_amdgcn_s_memrealtime();
#endif
    const long long u0 = sk_u0(r, sk), u1 = sk_u0(r + 1, sk);
    if (u0 >= u1) return;

    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, sk.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, sk.w_bytes, 0x00020000);
    const unsigned row_bytes = (unsigned)a.in_ch * 4u;
    const unsigned ring_bytes = (unsigned)a.in_rows * row_bytes;
    const unsigned dil_bytes = (unsigned)a.dilation * row_bytes;
    const unsigned lane16 = (unsigned)lane * 16u;
    constexpr unsigned OOB = 0x80000000u;

    // ---- activation staging state: describes the NEXT chunk whose X columns are to be loaded (XPD chunks ahead of the MFMAs) ----
    int s_tile, s_kc;                                  // wave-uniform
    int s_g = 0, s_mt = 0, s_nt = 0;
    int t_tap, t_cblk;                                 // per thread: tap / 32-channel block of ITS half-chunk
    unsigned colb[RB], rowb[RB];                       // per staged column: stream+channel base, ring row (tap applied)

    auto stage_tile = [&](int tile, int kc0) {
        sk_tile_coords(tile, sk, s_g, s_mt, s_nt);
        s_tile = tile; s_kc = kc0;
        const int j = 2 * KD * kc0 + half;             // index of this thread's 32-deep sub-chunk
        t_tap = j / sk.cpt; t_cblk = j - t_tap * sk.cpt;
        const int n0 = s_nt * BN;
        const unsigned tap_bytes = (unsigned)t_tap * dil_bytes;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const int n = n0 + srow + CPR * rr;
            const int nn = n < a.n_total ? n : 0;
            const int b = fast_div(nn, a.t_out, sk.inv_t_out), t = nn - b * a.t_out;
            int row = a.in_row0 + t * a.stride;
            if (row >= a.in_rows) row -= a.in_rows;
            unsigned rbv = (unsigned)row * row_bytes + tap_bytes;
            if (rbv >= ring_bytes) rbv -= ring_bytes;
            rowb[rr] = rbv;
            colb[rr] = n < a.n_total ? ((unsigned)b * ring_bytes + (unsigned)(a.in_choff + s_g * a.in_gstride + 4 * (quad & 7)) * 4u) : OOB;
        }
    };
    auto stage_advance = [&]() {                       // staged chunk -> next chunk (maybe next tile)
        ++s_kc;
        if (s_kc == sk.nchunks) { stage_tile(s_tile + 1, 0); return; }
        t_cblk += 2 * KD;                              // this thread's sub-chunk moves on by one chunk of 32-blocks
        while (t_cblk >= sk.cpt) {
            t_cblk -= sk.cpt; ++t_tap;
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {
                unsigned rbv = rowb[rr] + dil_bytes;
                if (rbv >= ring_bytes) rbv -= ring_bytes;
                rowb[rr] = rbv;
            }
        }
    };
    // ---- weight stream state: the NEXT chunk whose fragments are to be loaded (always one chunk ahead) ----
    int w_tile, w_kc;
    unsigned w_base = 0;                               // byte offset of this wave's fragment stream (out of bounds if it has none)
    auto w_set = [&](int tile, int kc0) {
        int g, mt, nt;
        sk_tile_coords(tile, sk, g, mt, nt);
        w_tile = tile; w_kc = kc0;
        const int mtile32 = mt * WGM + wm;             // 32-row fragment tile inside the group
        w_base = mtile32 < sk.mt32_per_g ? (unsigned)((g * sk.mt32_per_g + mtile32) * sk.kgroups) * 1024u : 0xfff00000u;
    };
    auto w_advance = [&]() { if (++w_kc == sk.nchunks) w_set(w_tile + 1, 0); };

    // Two register sets for each operand, used alternately by even and odd iterations (the loop below is unrolled by two with
    // the roles swapped): nothing is ever copied, so a load is only waited for where its data is consumed -- a weight
    // fragment at the MFMAs of the NEXT iteration, an activation piece when it is converted into LDS.  (Round 1 shifted
    // `next` into `current` registers at the end of every iteration, which made every iteration end in s_waitcnt vmcnt(0).)
    // XPD = 2 (split variant, 64-deep chunks): the activation loads run TWO chunks ahead -- issued at the top of iteration i,
    // converted at the bottom of iteration i + 1 -- so they, too, have a whole iteration to land.
    constexpr int XPD = (SPLIT && KD == 1 && NJ <= 2) ? 2 : 1;
#ifndef ADK_SK16_ILV
#define ADK_SK16_ILV 1      // 1: loads, MFMAs and conversion of an iteration interleaved in a pinned order (0: three phases)
#endif
    constexpr bool ILV = XPD == 2 && ADK_SK16_ILV > 0;
    float4 rbuf[2][RB];
    float4 abuf[2][8 * KD];
    // `valid` = false: every load goes out of bounds (returns 0 without touching memory) -- lets the last iterations run the
    // same straight-line code as the others
    auto gloadX = [&](float4 (&rb)[RB], bool valid = true) {
        const bool k_ok = valid && t_tap < a.taps;     // false on the zero-padded K tail
        const unsigned cb = (unsigned)t_cblk * 128u;
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            const unsigned vo = (k_ok && colb[rr] != OOB) ? colb[rr] + rowb[rr] + cb : OOB;
            rb[rr] = buf_load4(rsrc_in, vo, 0);
        }
    };
    auto gloadW = [&](float4 (&af)[8 * KD], bool valid = true) {
        const unsigned sa = valid ? w_base + (unsigned)w_kc * (8192u * KD) : 0xfff00000u;
#pragma unroll
        for (int q = 0; q < 8 * KD; ++q) af[q] = buf_load4(rsrc_w, lane16, sa + q * 1024u);
    };
    auto lstore_piece = [&](int buf, const float4 (&rb)[RB], int rr) {
        float* Bb = Bs + buf * BN * LDK;
        if constexpr (SPLIT && ACT == kActPre) {
            // shadow ring: this thread's 16 bytes ARE 8 halfs of the operand form -- the hi halfs of an 8-channel group (even piece) or
            // its lo halfs (odd piece): one 16-byte LDS store, no arithmetic
            unsigned char* d = reinterpret_cast<unsigned char*>(Bb + (srow + CPR * rr) * LDK) + 16 * (quad >> 1) + ((quad & 1) ? 2 * KCC : 0);
            *reinterpret_cast<float4*>(d) = rb[rr];
        } else {
            float4 v = rb[rr];
            if (!(SPLIT && (ADK_SK16_DBG & 2))) {
                v.x = act_in_apply<ACT>(v.x, a.slope); v.y = act_in_apply<ACT>(v.y, a.slope);
                v.z = act_in_apply<ACT>(v.z, a.slope); v.w = act_in_apply<ACT>(v.w, a.slope);
            }
            if constexpr (SPLIT) {
                // column row = [KCC halfs hi][KCC halfs lo][16 B pad]; this thread's 4 floats -> 2 x 8 bytes
                f16x4s hi, lo;
                const float x[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const _Float16 h = (_Float16)x[e];
                    hi[e] = h;
                    lo[e] = (ADK_SK16_DBG & 1) ? (_Float16)0.f : (_Float16)((x[e] - (float)h) * kSkLoScale);
                }
                unsigned char* d = reinterpret_cast<unsigned char*>(Bb + (srow + CPR * rr) * LDK) + 8 * quad;
                *reinterpret_cast<f16x4s*>(d) = hi;
                *reinterpret_cast<f16x4s*>(d + 2 * KCC) = lo;
            } else {
                *reinterpret_cast<float4*>(Bb + (srow + CPR * rr) * LDK + 4 * quad) = v;
            }
        }
    };
    auto lstore = [&](int buf, const float4 (&rb)[RB]) {
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) lstore_piece(buf, rb, rr);
    };

    f32x16 acc[NJ];
    f32x16 accx[SPLIT ? NJ : 1];                       // SPLIT: cross-term accumulators (scaled by 2048)
#pragma unroll
    for (int j = 0; j < NJ; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
#pragma unroll
    for (int j = 0; j < (SPLIT ? NJ : 1); ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) accx[j][e] = 0.f;

    // ---- prologue: first chunk into LDS buffer 0 / register set 0 ----
    int tile = (int)(u0 / sk.nchunks);
    int kc = (int)(u0 - (long long)tile * sk.nchunks);
    int seg_start_kc = kc;                             // first chunk of the current segment
    const int n_units = (int)(u1 - u0);
    stage_tile(tile, kc);
    w_set(tile, kc);
    int cur_g = s_g, cur_mt = s_mt, cur_nt = s_nt;
    gloadX(rbuf[0]);
    if constexpr (ILV) {
        // chunk 0's weights and chunk 1's activations in exactly the order (and registers) an odd iteration issues them, so that
        // the compiler's s_waitcnt bookkeeping at the loop head is the same from here as from the back edge (it merges the two
        // conservatively: with another order the first even iteration of every pair drained all loads)
        constexpr int RPS = RB / 4;
        const bool v1 = n_units > 1;
        if (v1) stage_advance();
        const unsigned sa = w_base + (unsigned)w_kc * 8192u;
        const bool k_ok = v1 && t_tap < a.taps;
        const unsigned cb = (unsigned)t_cblk * 128u;
#pragma unroll
        for (int st = 0; st < 4; ++st) {
            abuf[0][2 * st] = buf_load4(rsrc_w, lane16, sa + (2 * st) * 1024u);
            abuf[0][2 * st + 1] = buf_load4(rsrc_w, lane16, sa + (2 * st + 1) * 1024u);
#pragma unroll
            for (int i = 0; i < RPS; ++i) {
                const int rr = st * RPS + i;
                const unsigned vo = (k_ok && colb[rr] != OOB) ? colb[rr] + rowb[rr] + cb : OOB;
                rbuf[1][rr] = buf_load4(rsrc_in, vo, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    } else {
        gloadW(abuf[0]);
        if (XPD == 2 && n_units > 1) { stage_advance(); gloadX(rbuf[1]); }     // chunk 1's activations: consumed at the bottom of iteration 0
    }
    lstore(0, rbuf[0]);
    __syncthreads();

#if ADK_SK16_DBG & 16
    __shared__ unsigned long long trace_lds[64 * 8];
    const bool trace_on = (r == sk.G / 2) && wave == 0;
    if (trace_on) for (int i = lane; i < 64 * 8; i += 64) trace_lds[i] = 0ull;
#endif
    auto iteration = [&](auto parity, int it) {
        constexpr int P = decltype(parity)::value;
        constexpr int cur = P;
        SK_STAMP(0);
        // -- issue the loads of the coming chunks (possibly of the next tile) --
        const bool has_next = (it + 1 < n_units);
        if constexpr (XPD == 2) {
            // address updates first (they may branch), then ONE straight-line block of loads, MFMAs and conversion that the
            // scheduler is told to interleave (below): measured with s_memtime stamps (profiles/r2_sk16_timeline.md), the
            // round-1 order -- all loads, all MFMAs, all conversions, barrier -- ran as three serial phases of ~1000 / 600 /
            // 1000 cycles in which the four waves of a workgroup, in lockstep behind the barrier, queue up at the same unit
            const bool has_next2 = (it + 2 < n_units);
            if (has_next) w_advance();
            if (has_next2) stage_advance();
            if (!ILV) { gloadW(abuf[P ^ 1], has_next); gloadX(rbuf[P], has_next2); }
        } else {
            if (has_next) { w_advance(); gloadW(abuf[P ^ 1]); stage_advance(); gloadX(rbuf[P ^ 1]); }
        }
        const float4 (&a_cur)[8 * KD] = abuf[P];
        if (!ILV) SK_STAMP(1);
        // -- MFMAs on the current chunk --
        const float* Bb = Bs + cur * BN * LDK + (wn * NJ * 32 + l31) * LDK + 4 * lh;
        if constexpr (ILV) {
            // One straight-line block, order pinned by sched_barrier(0): per 16-k step -- B fragments, the two MFMAs that start
            // the accumulator chains, a quarter of the conversion of the NEXT chunk's activations into the other LDS buffer
            // (VALU work that runs under those MFMAs), the third MFMA, then a quarter of this iteration's loads (weights of the
            // next chunk, activations of the one after).  The four waves of a workgroup thus reach the vector-memory unit, the
            // matrix cores and the VALU at different times instead of queueing up at each in turn.
            static_assert(RB % 4 == 0 && KD == 1, "interleaved variant: 4 k-steps, RB / 4 staging pieces each");
            constexpr int RPS = RB / 4;
            const bool has_next2 = (it + 2 < n_units);
            const unsigned char* Bh = reinterpret_cast<const unsigned char*>(Bb);
#if ADK_SK16_DBG & 64
            // knock-out (results are garbage): the waves with wn == 1 request no weight fragments at all -- what would the launch cost if the two
            // waves of a workgroup that multiply the SAME 32 rows shared one copy of their A fragments (half the A traffic)?
            const unsigned sa = (has_next && wn == 0) ? w_base + (unsigned)w_kc * 8192u : 0xfff00000u;
#else
            const unsigned sa = has_next ? w_base + (unsigned)w_kc * 8192u : 0xfff00000u;
#endif
            const bool k_ok = has_next2 && t_tap < a.taps;
            const unsigned cb = (unsigned)t_cblk * 128u;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int st = 0; st < 4; ++st) {
                union { float4 f; f16x8s h; } ah, al;
                ah.f = a_cur[2 * st]; al.f = a_cur[2 * st + 1];
                f16x8s bh[NJ], bl[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8s*>(Bh + j * 32 * LDK * 4 + 32 * st);
                    bl[j] = *reinterpret_cast<const f16x8s*>(Bh + j * 32 * LDK * 4 + 32 * st + 2 * KCC);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bl[j], accx[j], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int i = 0; i < RPS; ++i) lstore_piece(cur ^ 1, rbuf[P ^ 1], st * RPS + i);   // (zeros after the last chunk: never read)
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, bh[j], accx[j], 0, 0, 0);
                abuf[P ^ 1][2 * st] = buf_load4(rsrc_w, lane16, sa + (2 * st) * 1024u);
                abuf[P ^ 1][2 * st + 1] = buf_load4(rsrc_w, lane16, sa + (2 * st + 1) * 1024u);
#pragma unroll
                for (int i = 0; i < RPS; ++i) {
                    const int rr = st * RPS + i;
                    const unsigned vo = (k_ok && colb[rr] != OOB) ? colb[rr] + rowb[rr] + cb : OOB;
                    rbuf[P][rr] = buf_load4(rsrc_in, vo, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        } else if constexpr (SPLIT) {
            const unsigned char* Bh = reinterpret_cast<const unsigned char*>(Bb);     // + 4*lh floats = 16*lh bytes: this lane's 8 halfs
#pragma unroll
            for (int st = 0; st < 4 * KD; ++st) {
                union { float4 f; f16x8s h; } ah, al;
                ah.f = a_cur[2 * st]; al.f = a_cur[2 * st + 1];
                f16x8s bh[NJ], bl[NJ];
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    bh[j] = *reinterpret_cast<const f16x8s*>(Bh + j * 32 * LDK * 4 + 32 * st);
                    bl[j] = *reinterpret_cast<const f16x8s*>(Bh + j * 32 * LDK * 4 + 32 * st + 2 * KCC);
                }
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh[j], acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bl[j], accx[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) accx[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, bh[j], accx[j], 0, 0, 0);
                if (st == 0 && !ILV) SK_STAMP(2);
            }
        } else {
            float4 bv[2][NJ];
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[0][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK);
#pragma unroll
            for (int q = 0; q < 8; ++q) {
                if (q + 1 < 8) {
#pragma unroll
                    for (int j = 0; j < NJ; ++j) bv[(q + 1) & 1][j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + (q + 1) * 8);
                }
                const float4 av = a_cur[q];
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv[q & 1][j].x, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv[q & 1][j].y, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, bv[q & 1][j].z, acc[j], 0, 0, 0);
#pragma unroll
                for (int j = 0; j < NJ; ++j) acc[j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, bv[q & 1][j].w, acc[j], 0, 0, 0);
            }
        }
        if (!ILV) SK_STAMP(3);
        // -- stage the next chunk --
        if constexpr (XPD == 2) {
            if (!ILV) lstore(cur ^ 1, rbuf[P ^ 1]);     // (zeros after the last chunk: nobody reads that buffer again)
        } else {
            if (has_next) lstore(cur ^ 1, rbuf[P ^ 1]);
        }
        SK_STAMP(4);
        // -- end of this tile's segment? --
        if (kc == sk.nchunks - 1 || !has_next) {
            const int ml0 = (cur_mt * WGM + wm) * 32;
            const int n0w = cur_nt * BN + wn * NJ * 32;
            if constexpr (SPLIT) {
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) { acc[j][e] = fmaf(accx[j][e], kSkLoInv, acc[j][e]); accx[j][e] = 0.f; }
            }
            const bool seg_first = (seg_start_kc == 0);            // segment holds the tile's first chunk
            const bool seg_last = (kc == sk.nchunks - 1);          // ... and its last chunk
            if (!seg_first) {
                // Head of this range: the tile started in an earlier range, whose workgroup owns it.
                // Publish the raw partial accumulators: write-through (sc1) stores, drain, one flag.
                const __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc(sk.ws, 0, sk.ws_bytes, 0x00020000);
                // layout [range][16-byte piece q][thread]: a wave's store of one piece is 1 KiB contiguous (with the thread-major layout of
                // rounds 1-3 its 64 lanes were 64 * NJ bytes apart: four times the cache lines per store, and per load on the owner's side)
                const unsigned wbase = (unsigned)r * (unsigned)(NT * NJ * 64) + (unsigned)tid * 16u;
#pragma unroll
                for (int j = 0; j < NJ; ++j)
#pragma unroll
                    for (int e4 = 0; e4 < 4; ++e4) {
                        u32x4 v;
                        v.x = __float_as_uint(acc[j][4 * e4]); v.y = __float_as_uint(acc[j][4 * e4 + 1]);
                        v.z = __float_as_uint(acc[j][4 * e4 + 2]); v.w = __float_as_uint(acc[j][4 * e4 + 3]);
                        __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_ws, wbase + (unsigned)((j * 4 + e4) * (NT * 16)), 0, 16 /* sc1 */);
                    }
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // every storing wave drains
                __syncthreads();
                if (tid == 0) __hip_atomic_store(sk.flags + r, sk.epoch, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
                if (!seg_last) {
                    // Owner of a tile that continues in the following range(s): their head partials were
                    // produced at the START of those workgroups' runs, this is the END of ours.  Add them in
                    // range order (deterministic): one relaxed poll loop + one agent acquire per contributor.
                    const long long t1 = ((long long)tile + 1) * sk.nchunks;
                    int rr_end = r + 1;
                    while (rr_end < sk.G && sk_u0(rr_end, sk) < t1) ++rr_end;
                    if (tid == 0) {                                // wait for ALL contributors, then one acquire
                        for (int rr = r + 1; rr < rr_end; ++rr) {
                            if (sk_u0(rr + 1, sk) <= sk_u0(rr, sk)) continue;   // empty range: publishes nothing
                            unsigned spins = 0;
                            while (__hip_atomic_load(sk.flags + rr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != sk.epoch) {
                                __builtin_amdgcn_s_sleep(2);
                                if (++spins > (1u << 20)) { atomicOr(sk.err, 2); break; }     // never hang the device
                            }
                        }
#if !ADK_SK_SC1_READ
                        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
#endif
                    }
                    __syncthreads();
#if ADK_SK_SC1_READ
                    // the partial tiles were stored write-through (sc1); reading them with sc1 loads (served by L2, never by this
                    // CU's L1) needs no acquire fence (MI355X_MICROARCH.md, inter-workgroup visibility: "16 B sc1 stores AND sc1
                    // loads"; the fence costs ~1.7 us)
                    const __amdgpu_buffer_rsrc_t rsrc_rd = __builtin_amdgcn_make_buffer_rsrc(sk.ws, 0, sk.ws_bytes, 0x00020000);
#endif
                    for (int rr = r + 1; rr < rr_end; ++rr) {
                        if (sk_u0(rr + 1, sk) <= sk_u0(rr, sk)) continue;
#if ADK_SK_SC1_READ
                        const unsigned rbase = (unsigned)rr * (unsigned)(NT * NJ * 64) + (unsigned)tid * 16u;
                        u32x4 pv[NJ * 4];
#pragma unroll
                        for (int q = 0; q < NJ * 4; ++q) pv[q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_rd, rbase + (unsigned)(q * (NT * 16)), 0, 16 /* sc1 */);
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const u32x4 v = pv[j * 4 + e4];
                                acc[j][4 * e4] += __uint_as_float(v.x); acc[j][4 * e4 + 1] += __uint_as_float(v.y);
                                acc[j][4 * e4 + 2] += __uint_as_float(v.z); acc[j][4 * e4 + 3] += __uint_as_float(v.w);
                            }
#else
                        const float* wsp = sk.ws + (size_t)rr * NT * (NJ * 16) + (size_t)tid * 4;
#pragma unroll
                        for (int j = 0; j < NJ; ++j)
#pragma unroll
                            for (int e4 = 0; e4 < 4; ++e4) {
                                const float4 v = *reinterpret_cast<const float4*>(wsp + (size_t)(j * 4 + e4) * (NT * 4));
                                acc[j][4 * e4] += v.x; acc[j][4 * e4 + 1] += v.y; acc[j][4 * e4 + 2] += v.z; acc[j][4 * e4 + 3] += v.w;
                            }
#endif
                    }
                }
                if constexpr (WGM <= 2) {
                    if (sk.epi_lds) {
                        // buffer `cur` was this iteration's B operand: free once every wave is past its MFMAs; the next iteration
                        // stores into it again only behind the barrier at the end of this one
                        __syncthreads();
                        sk_epilogue_lds<NJ, 32 * WGM, BN, NT, SPLIT>(a, acc, Bs + cur * BN * LDK, cur_g, cur_mt * (32 * WGM), cur_nt * BN, wm, wn, tid, sk.inv_t_out, sk.err);
                    } else {
                        sk_epilogue<NJ, SPLIT>(a, acc, cur_g, ml0, n0w, lane, sk.err);
                    }
                } else {
                    sk_epilogue<NJ, SPLIT>(a, acc, cur_g, ml0, n0w, lane, sk.err);
                }
            }
#pragma unroll
            for (int j = 0; j < NJ; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[j][e] = 0.f;
            seg_start_kc = 0;
            if (has_next) sk_tile_coords(tile + 1, sk, cur_g, cur_mt, cur_nt);
        }
        SK_STAMP(5);
        __syncthreads();
        SK_STAMP(6);
        if (++kc == sk.nchunks) { kc = 0; ++tile; }
    };
#if ADK_SK16_DBG & 32
    const unsigned long long wg_t1 = __builtin_amdgcn_s_memrealtime();
    unsigned long long wg_t2 = 0;
#endif
    for (int it = 0; it < n_units; it += 2) {
#if ADK_SK16_DBG & 32
        if (it + 2 >= n_units) wg_t2 = __builtin_amdgcn_s_memrealtime();      // before the last (pair of) iteration(s): the segment-end wait is in there
#endif
        iteration(std::integral_constant<int, 0>(), it);
        if (it + 1 < n_units) iteration(std::integral_constant<int, 1>(), it + 1);
    }
#if ADK_SK16_DBG & 32
    if (tid == 0 && r < 512) {
        g_sk_wg_trace[r * 4] = wg_t0; g_sk_wg_trace[r * 4 + 1] = wg_t1; g_sk_wg_trace[r * 4 + 2] = wg_t2;
        g_sk_wg_trace[r * 4 + 3] = __builtin_amdgcn_s_memrealtime();
    }
#endif
#if ADK_SK16_DBG & 16
    if (trace_on) for (int i = lane; i < 64 * 8; i += 64) g_sk_trace[i] = trace_lds[i];
#endif
}


// ================================================================================================
// conv_gv16 -- the split-f16 conv for FEW COLUMNS (a handful of streams, layers of 1-25 steps per frame; by default n_total <= 32, option
// "gv16_max_columns"), round 5.
//
// One frame of one stream needs every weight once (93 MB for vctk_v1) and multiplies it with a handful of columns: a GEMV.  The stream-K
// kernel above runs such a conv as 64 x 64 tiles whose K is split over at most five workgroups -- a dozen workgroups on a chip of 256 CUs, each
// walking 6-9 chunks through LDS staging and barriers, then a partial-tile exchange: 10-17 us per launch whatever the work
// (profiles/r5_single_stream_latency.md: 23 such launches are 43 % of a single-stream frame).  Here the conv is cut the other way:
//   * a work item = one 32-row m-tile x one 32-column n-tile x one slice of <= MAXS 16-k steps of K, ONE WAVE (64 threads) per item -- hundreds of waves, every CU
//     pulls a few KiB of weights, all of an item's loads (weights: 2 KiB per step; its 32 columns' operands: 2 x 16 B per lane and step,
//     straight from the state ring or its shadow) are requested up front, in straight-line code, and consumed in order: one round trip;
//   * no LDS, no barrier: the B fragment of a lane IS 8 consecutive channels of its column's ring row (activation + split in registers,
//     or the shadow's [8 hi][8 lo] group as it is);
//   * the S slices of a tile leave their 32 x 32 partial sums in the workspace (write-through), count in on the tile's counter, and the LAST
//     one to arrive adds all S in slice order -- a fixed order: the result does not depend on who was last -- and runs the stream-K
//     kernel's epilogue (bias, residual, output activation, shadow).  Nobody waits for anybody.
// Per accumulator the order is that of the other split kernels (hi*hi | hi*lo, lo*hi; main + cross / 2048 per slice); where K is cut differs
// from the stream-K kernel, so results agree with it to f32 round-off, not bit for bit; the same call is bit-reproducible.
struct GvArgs {
    float* ws; unsigned ws_bytes;        // partial sums: [item][4 pieces of 16 B][64 lanes]
    unsigned* counters;                  // [tiles][ntiles]: slices of the tile that have published; 0 between launches
    int S;                               // K slices per tile
    int tiles;                           // groups * mt32_per_g
    int ntiles;                          // 32-column tiles: (n_total + 31) / 32
    int mt32_per_g;
    int ksteps, ksteps_packed;           // 16-k steps of K (ktot / 16) / of the packed weights (K padded to 64)
    int cpt16;                           // 16-k steps per tap = cin_g / 16
    unsigned in_bytes, w_bytes;
    float inv_t_out;
    int* err;
};

template <int ACT, int MAXS>
__global__ __launch_bounds__(64) void conv_gv16_kernel(ConvArgs a, GvArgs gv) {
    const int lane = threadIdx.x, l31 = lane & 31, lh = lane >> 5;
    const int item = blockIdx.x;
    const int otile = item / gv.S, slice = item - otile * gv.S;          // otile = (m-tile, n-tile): slices of one output tile are neighbours
    const int tile = otile / gv.ntiles, nt = otile - tile * gv.ntiles;
    if (tile >= gv.tiles) return;
    const int g = tile / gv.mt32_per_g, mt = tile - g * gv.mt32_per_g;
    const int s0 = (int)(((long long)slice * gv.ksteps) / gv.S), s1 = (int)(((long long)(slice + 1) * gv.ksteps) / gv.S);
    constexpr unsigned OOB = 0x80000000u;
    const __amdgpu_buffer_rsrc_t rsrc_in = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.in), 0, gv.in_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t rsrc_w = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(a.wfrag), 0, gv.w_bytes, 0x00020000);
    const unsigned lane16 = (unsigned)lane * 16u;
    // this lane's column (stream b, step t): ring row of tap 0, channel block of its k-half
    const bool col_ok = nt * 32 + l31 < a.n_total;
    const int nn = col_ok ? nt * 32 + l31 : 0;
    const int b = fast_div(nn, a.t_out, gv.inv_t_out), t = nn - b * a.t_out;
    const unsigned row_bytes = (unsigned)a.in_ch * 4u, ring_bytes = (unsigned)a.in_rows * row_bytes, dil_bytes = (unsigned)a.dilation * row_bytes;
    int row0 = a.in_row0 + t * a.stride;
    if (row0 >= a.in_rows) row0 -= a.in_rows;
    const unsigned colb = col_ok ? (unsigned)b * ring_bytes + (unsigned)(a.in_choff + g * a.in_gstride + 8 * lh) * 4u : OOB;
    const unsigned wbase = (unsigned)((g * gv.mt32_per_g + mt) * gv.ksteps_packed) * 2048u;
    int tap = s0 / gv.cpt16, c16 = s0 - tap * gv.cpt16;
    unsigned rowb = (unsigned)row0 * row_bytes + (unsigned)tap * dil_bytes;
    if (rowb >= ring_bytes) rowb -= ring_bytes;

    // ---- every load of the item, up front (steps past the slice's end go out of bounds: zeros, no memory touched) ----
    u32x4 ah[MAXS], al[MAXS];
    float4 x0[MAXS], x1[MAXS];
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        const bool ok = s0 + i < s1;
        const unsigned wo = ok ? wbase + (unsigned)(s0 + i) * 2048u : 0xfff00000u;
        ah[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16, wo, 0);
        al[i] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_w, lane16 + 1024u, wo, 0);
        const unsigned vo = (ok && col_ok) ? colb + rowb + (unsigned)c16 * 64u : OOB;
        x0[i] = buf_load4(rsrc_in, vo, 0);
        x1[i] = buf_load4(rsrc_in, vo, 16);
        if (++c16 == gv.cpt16) {
            c16 = 0;
            rowb += dil_bytes;
            if (rowb >= ring_bytes) rowb -= ring_bytes;
        }
    }
    __builtin_amdgcn_sched_barrier(0);          // (keep the requests HERE: left alone the scheduler sinks each load to just in front of its use)

    f32x16 acc[1], accx;
#pragma unroll
    for (int e = 0; e < 16; ++e) { acc[0][e] = 0.f; accx[e] = 0.f; }
#pragma unroll
    for (int i = 0; i < MAXS; ++i) {
        union { u32x4 u; f16x8s h; } Ah, Al;
        Ah.u = ah[i]; Al.u = al[i];
        f16x8s bh, bl;
        if constexpr (ACT == kActPre) {
            union { float4 f; f16x8s h; } c0, c1;
            c0.f = x0[i]; c1.f = x1[i];
            bh = c0.h; bl = c1.h;
        } else {
            const float x[8] = {x0[i].x, x0[i].y, x0[i].z, x0[i].w, x1[i].x, x1[i].y, x1[i].z, x1[i].w};
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = act_in_apply<ACT>(x[e], a.slope);
                const _Float16 h = (_Float16)y;
                bh[e] = h;
                bl[e] = (_Float16)((y - (float)h) * kSkLoScale);
            }
        }
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah.h, bh, acc[0], 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah.h, bl, accx, 0, 0, 0);
        accx = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al.h, bh, accx, 0, 0, 0);
    }
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[0][e] = fmaf(accx[e], kSkLoInv, acc[0][e]);

    if (gv.S > 1) {
        // publish this slice's partial sums (write-through), count in; the last slice to arrive adds all of them in slice order
        const __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc(gv.ws, 0, gv.ws_bytes, 0x00020000);
        const unsigned pbase = (unsigned)item * 4096u + lane16;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            u32x4 v;
            v.x = __float_as_uint(acc[0][4 * q]); v.y = __float_as_uint(acc[0][4 * q + 1]);
            v.z = __float_as_uint(acc[0][4 * q + 2]); v.w = __float_as_uint(acc[0][4 * q + 3]);
            __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_ws, pbase + (unsigned)q * 1024u, 0, 16 /* sc1 */);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        unsigned old = 0;
        if (lane == 0) old = __hip_atomic_fetch_add(gv.counters + otile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        old = __builtin_amdgcn_readfirstlane(old);
        if (old != (unsigned)(gv.S - 1)) return;
        if (lane == 0) __hip_atomic_store(gv.counters + otile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // (the slices were stored write-through; sc1 loads are served past this CU's L1: no acquire fence -- as the stream-K kernel's owners)
        u32x4 pv[15][4];
        const unsigned rb0 = (unsigned)(otile * gv.S) * 4096u + lane16;
#pragma unroll
        for (int sl = 0; sl < 15; ++sl) {
            const unsigned ro = sl < gv.S ? rb0 + (unsigned)sl * 4096u : OOB;
#pragma unroll
            for (int q = 0; q < 4; ++q) pv[sl][q] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, ro, (unsigned)q * 1024u, 16 /* sc1 */);
        }
#pragma unroll
        for (int e = 0; e < 16; ++e) acc[0][e] = 0.f;
#pragma unroll
        for (int sl = 0; sl < 15; ++sl) {
            if (sl < gv.S) {
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc[0][4 * q] += __uint_as_float(pv[sl][q].x); acc[0][4 * q + 1] += __uint_as_float(pv[sl][q].y);
                    acc[0][4 * q + 2] += __uint_as_float(pv[sl][q].z); acc[0][4 * q + 3] += __uint_as_float(pv[sl][q].w);
                }
            }
        }
    }
    sk_epilogue<1, true>(a, acc, g, mt * 32, nt * 32, lane, gv.err);
}


// fragment packing: w [groups*cout_g][ktot] row-major -> [g][m-tile32][k-group8][lane64][4]
// lane (i = lane&31, h = lane>>5) holds W[32*mt + i][8*kg + 4*h + 0..3]; rows >= cout_g and the K
// tail (K is padded to a multiple of 64) are zero.
__global__ void pack_weights_kernel(const float* __restrict__ w, float* __restrict__ out, int groups, int cout_g, int ktot) {
    const int mt32 = (cout_g + 31) / 32, kg = (ktot + 63) / 64 * 8;
    const long long total = (long long)groups * mt32 * kg * 256;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long long)gridDim.x * blockDim.x) {
        const int e = (int)(i & 3), lane = (int)((i >> 2) & 63);
        long long rest = i >> 8;
        const int kgi = (int)(rest % kg); rest /= kg;
        const int mt = (int)(rest % mt32); const int g = (int)(rest / mt32);
        const int row = 32 * mt + (lane & 31);
        const int k = 8 * kgi + 4 * (lane >> 5) + e;
        out[i] = (row < cout_g && k < ktot) ? w[((size_t)g * cout_g + row) * ktot + k] : 0.f;
    }
}

bool conv_mfma_supported(const ConvArgs& a) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!a.wfrag) return false;
    // 32-bit buffer addressing: input arena view < 2 GiB, packed weights < 4 GiB, N < 2^24
    if ((unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull >= 0x80000000ull) return false;
    if ((unsigned long long)a.groups * ((a.cout_g + 31) / 32) * ((a.ktot + 63) / 64 * 8) * 1024ull >= 0xfff00000ull) return false;
    if (a.n_total >= (1 << 24)) return false;
    if (a.cin_g % 32 != 0 || a.cout_g % 4 != 0 || a.cout_real % 4 != 0) return false;
    if (a.in_ch % 4 || a.in_choff % 4 || a.in_gstride % 4 || a.out_ch % 4 || a.out_choff % 4) return false;
    if (!al16(a.in) || !al16(a.out) || !al16(a.wfrag) || (a.bias && !al16(a.bias))) return false;
    if (a.res && (a.res_ch % 4 || a.res_choff % 4 || a.res_gstride % 4 || !al16(a.res))) return false;
    return true;
}

// ------------------------------------------------------------------------------------------------
namespace {
struct Cfg { int wgm, wgn, nj; const char* name; };
const Cfg kCfgs[7] = {{4, 1, 2, "conv_sk<128x64>"}, {4, 1, 4, "conv_sk<128x128>"}, {2, 2, 1, "conv_sk<64x64>"},
                      {2, 2, 2, "conv_sk<64x128>"}, {1, 4, 1, "conv_sk<32x128>"}, {1, 4, 2, "conv_sk<32x256>"},
                      {4, 2, 2, "conv_sk<128x128w8>"}};       // 6: 8 waves (512 threads), split variant only
int g_forced_cfg = -2;     // -2: not initialised (read ADK_CONV_CFG), -1: heuristic
int g_occ = -1;            // persistent workgroups per CU (ADK_CONV_OCC, default 2)
int g_fixed_g = 0;         // persistent workgroups per launch when > 0 (ADK_CONV_G / adk_set_conv_workgroups), else 256 * g_occ
int g_oversub = 0;         // ADK_CONV_OVERSUB
int g_aligned = 1;         // tile-aligned stream-K ranges where the rule in launch_cfg applies (ADK_CONV_ALIGNED=0: never)
int g_max_split = 5;       // most workgroups sharing one tile (ADK_CONV_MAX_SPLIT; 0 = no limit).  Measured (tools/run_r2s.sh):
                           // 5 vs no limit at 256 streams: last strided conv 30.9 -> 18.3 us, first transposed conv 25.2 -> 18.9,
                           // K10 strided 24.1 -> 19.3; at 1 stream the grouped K11 256-channel conv 28.3 -> 18.9, projector 19.9 -> 13.9
int g_min_units = 2;       // minimum K chunks per workgroup (ADK_CONV_MIN_UNITS; 2 measured best at 256 streams)

// How many persistent workgroups a launch over `tiles` tiles of sk.nchunks chunks takes (at most `cap`) and where their ranges
// are cut: fills sk.split / tpw / tiles / owner_chunks / total, returns G.  Pure host logic (adk_streamk_plan exposes it to the
// CPU tests).
long long sk_plan(long long tiles, long long cap, SkArgs& sk) {
    sk.total = tiles * sk.nchunks;
    // persistent workgroups: 256 CUs x occupancy, but never fewer than g_min_units chunks per workgroup
    // (each one pays a fixed prologue/epilogue, and every cut of a tile costs a partial round trip)
    long long G = cap;
    const long long by_units = (sk.total + g_min_units - 1) / g_min_units;
    if (G > by_units) G = (by_units + 7) / 8 * 8;
    // ... and never more than g_max_split workgroups on one tile: its owner adds the others' partial tiles one after the
    // other, which is what a launch over few tiles (few streams, or the deepest layers) otherwise spends its time on
    if (g_max_split > 0 && G > tiles * g_max_split) G = (tiles * g_max_split + 7) / 8 * 8;
    sk.split = 0; sk.tpw = 0; sk.tiles = (int)tiles; sk.owner_chunks = 0;
    // Tile-aligned ranges (measured, tools/run_r2z.sh at 256 streams: a range that straddles two tiles pays two partial round
    // trips -- 200-tile 1x1 / strided / transposed convs 20.7 -> 10.8 / 13.0 / 13.4 us with one whole tile per workgroup, the
    // grouped K11 256-channel conv 50.8 -> 44.3 us with exact halves).  Short K (<= 8 chunks): whole tiles, as many per
    // workgroup as it takes.  Long K: an even split when >= 2 workgroups per tile fit (one workgroup per CU first, then
    // ~6 chunks each); a long-K layer with more tiles than that keeps the balanced split above.
    if (g_aligned && tiles < (1 << 20)) {
        const int ms = g_max_split > 0 ? g_max_split : 5;
        if (tiles > cap) {
            if (sk.nchunks <= 8) { sk.tpw = (int)((tiles + cap - 1) / cap); G = (tiles + sk.tpw - 1) / sk.tpw; }
        } else {
            long long sp = std::max<long long>(std::min<long long>(ms, 256 / tiles), std::min<long long>(ms, sk.nchunks / 6));
            if (sp < 1) sp = 1;
            while (sp > 1 && (tiles * sp > cap || sk.nchunks / sp < g_min_units)) --sp;
            if (sp > 1 || sk.nchunks < 12) { sk.split = (int)sp; G = tiles * sp; }
            // two workgroups per tile AND per CU: the owners run on the later-dispatched blocks (see the kernel), which get the
            // smaller share of a CU -- give them the smaller share of the tile (ADK_CONV_OWNER_SHARE percent, default 45)
            sk.owner_chunks = sk.nchunks / 2;
            if (sk.split == 2 && G >= 448) {             // (measured: grouped K11 256-channel conv, G = 480: 41.9 -> 39.5 us; G = 400 loses)
                static int share = -1;
                if (share < 0) { const char* e = getenv("ADK_CONV_OWNER_SHARE"); share = (e && atoi(e) >= 10 && atoi(e) <= 90) ? atoi(e) : 45; }
                sk.owner_chunks = std::max(g_min_units, std::min(sk.nchunks - g_min_units, (sk.nchunks * share + 50) / 100));
            }
        }
    }
    return G;
}

template <int WGM, int WGN, int NJ, bool SPLIT, int KD = 1>
int launch_cfg(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    constexpr int BM = 32 * WGM, BN = 32 * NJ * WGN, NT = 64 * WGM * WGN;
    constexpr int KCC = KC * KD;
    constexpr size_t lds = 2ull * BN * (KCC + 4) * sizeof(float);
    SkArgs sk;
    sk.m_tiles = (a.cout_g + BM - 1) / BM;
    sk.n_tiles = (a.n_total + BN - 1) / BN;
    sk.nchunks = (a.ktot + KCC - 1) / KCC;
    sk.cpt = a.cin_g / 32;
    sk.kgroups = (a.ktot + KC - 1) / KC * (KC / 8);   // packing stride: K padded to 64 whatever the chunk depth
    sk.mt32_per_g = (a.cout_g + 31) / 32;
    sk.inv_t_out = 1.0f / (float)a.t_out;
    {
        const unsigned long long inb = (unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull;
        const unsigned long long wb = (unsigned long long)a.groups * sk.mt32_per_g * sk.kgroups * 1024ull;
        if (inb >= 0x80000000ull || wb >= 0xfff00000ull || a.n_total >= (1 << 24))
            return fail(ADK_ERR_SHAPE, "conv: problem too large for the 32-bit buffer addressing of the MFMA kernel");
        sk.in_bytes = (unsigned)inb; sk.w_bytes = (unsigned)wb;
    }
    const long long tiles = (long long)sk.m_tiles * sk.n_tiles * a.groups;
    sk.total = tiles * sk.nchunks;
    const long long slots = NT == 256 ? 256LL * g_occ : 256LL;        // resident workgroups: 512-thread workgroups are alone on their CU
    // ADK_CONV_OVERSUB percent (tuning): allow that many more workgroups than slots -- the surplus starts when the first ones end
    const long long lim = NT == 256 ? slots + slots * g_oversub / 100 : slots;
    const long long cap = ws.workgroups > 0 ? std::min<long long>(ws.workgroups, lim) : (g_fixed_g > 0 ? std::min<long long>(g_fixed_g, lim) : lim);
    const long long G = sk_plan(tiles, cap, sk);
    sk.G = (int)G;
    const size_t part_bytes = (size_t)sk.G * NT * NJ * 16 * sizeof(float);
    if (!ws.ptr || part_bytes + (size_t)sk.G * sizeof(unsigned) > ws.bytes) return fail(ADK_ERR_STATE, "conv: stream-K workspace missing or too small");
    sk.ws = ws.ptr;
    sk.ws_bytes = (unsigned)part_bytes;
    sk.flags = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws.ptr) + ws.flags_offset);
    sk.epoch = ++ws.epoch;
    if (sk.epoch == 0) sk.epoch = ++ws.epoch;
    sk.err = conv_err_word(a);
    {
        // finished tiles through LDS (coalesced 16-byte accesses along the channel axis) where the tile image fits a staging buffer
        // (<= 64 rows) and 8-channel groups never straddle a row / a phase of a transposed conv; ADK_SK_EPI_LDS=0: the round-1..3 epilogue
        static int epi = -1;
        if (epi < 0) { const char* e = getenv("ADK_SK_EPI_LDS"); epi = e ? atoi(e) : 1; }
        sk.epi_lds = (epi && WGM <= 2 && a.cout_g % 8 == 0 && a.cout_real % 8 == 0 && a.out_ch % 8 == 0 && a.out_choff % 8 == 0 &&
                      (!a.res || (a.res_ch % 4 == 0 && a.res_choff % 4 == 0 && a.res_gstride % 4 == 0))) ? 1 : 0;
    }
    if (lds > 64 * 1024) {
        static bool attr_set_dev[kMaxDevices] = {};      // function attributes are per device
        bool& attr_set = attr_set_dev[current_device()];
        if (!attr_set) {
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sk_kernel<WGM, WGN, NJ, ADK_ACT_ELU, SPLIT, KD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sk_kernel<WGM, WGN, NJ, ADK_ACT_LEAKY, SPLIT, KD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sk_kernel<WGM, WGN, NJ, ADK_ACT_NONE, SPLIT, KD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
            attr_set = true;
        }
    }
    const unsigned grid = (unsigned)((sk.G + 7) / 8 * 8);
    if constexpr (SPLIT) {
        if (a.in_sh) {
            // the input ring has a shadow: stage the pre-activated, pre-split pieces from it (same geometry, same addressing)
            static bool pre_attr_dev[kMaxDevices] = {};
            bool& pre_attr = pre_attr_dev[current_device()];
            if (lds > 64 * 1024 && !pre_attr) {
                ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_sk_kernel<WGM, WGN, NJ, kActPre, SPLIT, KD>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
                pre_attr = true;
            }
            ConvArgs b = a;
            b.in = a.in_sh;
            hipLaunchKernelGGL((conv_sk_kernel<WGM, WGN, NJ, kActPre, SPLIT, KD>), dim3(grid), dim3(NT), lds, s, b, sk);
            ADK_HIP_CHECK(hipGetLastError());
            return ADK_OK;
        }
    }
    if (a.act_in == ADK_ACT_ELU)
        hipLaunchKernelGGL((conv_sk_kernel<WGM, WGN, NJ, ADK_ACT_ELU, SPLIT, KD>), dim3(grid), dim3(NT), lds, s, a, sk);
    else if (a.act_in == ADK_ACT_LEAKY)
        hipLaunchKernelGGL((conv_sk_kernel<WGM, WGN, NJ, ADK_ACT_LEAKY, SPLIT, KD>), dim3(grid), dim3(NT), lds, s, a, sk);
    else if (a.act_in == ADK_ACT_NONE)
        hipLaunchKernelGGL((conv_sk_kernel<WGM, WGN, NJ, ADK_ACT_NONE, SPLIT, KD>), dim3(grid), dim3(NT), lds, s, a, sk);
    else
        return fail(ADK_ERR_ARG, "conv: unsupported input activation for the MFMA kernel");
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}
}  // namespace

void conv_mfma_force_cfg(int cfg) { g_forced_cfg = cfg; }

int streamk_plan(long long tiles, int nchunks, int cap, int out[4]) {
    (void)conv_mfma_workspace_bytes(nullptr);                      // reads the env knobs
    SkArgs sk{};
    sk.nchunks = nchunks;
    const long long slots = 256LL * g_occ;
    const long long G = sk_plan(tiles, cap > 0 ? std::min<long long>(cap, slots) : slots, sk);
    out[0] = (int)G; out[1] = sk.split; out[2] = sk.tpw; out[3] = sk.owner_chunks;
    return ADK_OK;
}
long long streamk_range_start(long long tiles, int nchunks, const int plan[4], int r) {
    return sk_range_start(r, plan[0], plan[1], plan[2], (int)tiles, nchunks, plan[3], tiles * nchunks);
}

#if ADK_SK16_DBG & 32
extern "C" int adk_debug_sk_wg_trace(unsigned long long* out, int n) {   // debug builds only
    if (n > 512 * 4) n = 512 * 4;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sk_wg_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif
#if ADK_SK16_DBG & 16
extern "C" int adk_debug_sk_trace(unsigned long long* out, int n) {      // debug builds only (tools/kbench looks it up with dlsym)
    if (n > 64 * 8) n = 64 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_sk_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#endif

// workspace = partial slots [G][256][NJ<=4][16] floats, then G publish flags
size_t conv_mfma_workspace_bytes(size_t* flags_offset) {
    if (g_occ < 0) {
        const char* e = getenv("ADK_CONV_OCC"); g_occ = e ? atoi(e) : 2; if (g_occ < 1 || g_occ > 4) g_occ = 2;
        e = getenv("ADK_CONV_MIN_UNITS"); if (e && atoi(e) >= 1) g_min_units = atoi(e);
        e = getenv("ADK_CONV_MAX_SPLIT"); if (e && atoi(e) >= 0) g_max_split = atoi(e);
        e = getenv("ADK_CONV_ALIGNED"); if (e) g_aligned = atoi(e) != 0;
        e = getenv("ADK_CONV_OVERSUB"); if (e && atoi(e) >= 0 && atoi(e) <= 100) g_oversub = atoi(e);
        e = getenv("ADK_CONV_G"); if (e && atoi(e) >= 8 && atoi(e) <= 2 * 256 * g_occ) g_fixed_g = atoi(e) / 8 * 8;
    }
    const size_t part = (size_t)256 * g_occ * 256 * 4 * 16 * sizeof(float);
    if (flags_offset) *flags_offset = part;
    return part + (size_t)2 * 256 * g_occ * sizeof(unsigned)        // flags for up to twice the resident workgroups (oversubscribed plans)
           + (size_t)kGvCounters * sizeof(unsigned);                // ... and the per-tile arrival counters of conv_gv16 (zero between launches)
}

int conv_mfma_pick(const ConvArgs& a) {
    if (g_forced_cfg == -2) { const char* e = getenv("ADK_CONV_CFG"); g_forced_cfg = e ? atoi(e) : -1; }
    if (g_forced_cfg >= 0 && g_forced_cfg <= 5) return g_forced_cfg;
    // Measured over every layer of the path at 256 streams (profiles/r1_streamk_cfg_sweep.md): the
    // 32x32 wave tile wins everywhere -- 64x64 workgroup tiles when the group has >= 64 output
    // channels, 32x128 otherwise.  (The wider tiles are kept for tuning via ADK_CONV_CFG.)
    return (a.cout_g % 64 == 0) ? 2 : 4;
}

const char* conv_mfma_cfg_name(int pick) { return kCfgs[pick].name; }

int launch_conv_mfma(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    if (a.n_total == 0) return ADK_OK;
    (void)conv_mfma_workspace_bytes(nullptr);
    switch (conv_mfma_pick(a)) {
        case 0: return launch_cfg<4, 1, 2, false>(a, s, ws);
        case 1: return launch_cfg<4, 1, 4, false>(a, s, ws);
        case 2: return launch_cfg<2, 2, 1, false>(a, s, ws);
        case 3: return launch_cfg<2, 2, 2, false>(a, s, ws);
        case 4: return launch_cfg<1, 4, 1, false>(a, s, ws);
        default: return launch_cfg<1, 4, 2, false>(a, s, ws);
    }
}

// split-f16 variant: wfrag = adk_pack_weights_split16 layout
int conv_sk16_pick(const ConvArgs& a) {
    if (g_forced_cfg == -2) { const char* e = getenv("ADK_CONV_CFG"); g_forced_cfg = e ? atoi(e) : -1; }
    if (g_forced_cfg >= 0 && g_forced_cfg <= 6) return g_forced_cfg;
    // with the matrix-core time cut to 3/16 the weight / activation re-reads weigh more: 128-row tiles (each X chunk
    // staged once per 128 output channels) win when that still leaves >= 256 tiles (measured: grouped 128-channel
    // vocoder stage 47.7 vs 55.2 us; the 100-tile encoder block loses, 34.6 vs 28.2 us)
    static int wide = -1;                             // ADK_SK16_WIDE=1 (tuning): 128-row tiles whenever the group has >= 128 channels
    if (wide < 0) { const char* e = getenv("ADK_SK16_WIDE"); wide = e ? atoi(e) : 0; }
    if (a.cout_g % 128 == 0 && (wide || (long long)(a.cout_g / 128) * ((a.n_total + 63) / 64) * a.groups >= 256)) return 0;
    return (a.cout_g % 64 == 0) ? 2 : 4;
}


// ---- conv_gv16 host side ----
// (atomics + one call_once env read, as for the RVQ options: adk_set_option may be called from another host thread than the one launching)
static std::atomic<int> g_gv{1};         // ADK_GV16: 1 (default) = convs of few columns run as conv_gv16, 0 = never (the stream-K kernel takes them)
static std::atomic<int> g_gv_maxn{32};   // ... "few" = at most this many columns (ADK_GV16_MAXN / option "gv16_max_columns"; default 32 = one n-tile)
static std::once_flag g_gv_once;
static void gv_read_env() {
    std::call_once(g_gv_once, [] {
        const char* e = getenv("ADK_GV16"); if (e) g_gv.store(atoi(e));
        e = getenv("ADK_GV16_MAXN"); if (e) g_gv_maxn.store(atoi(e) < 0 ? 0 : atoi(e));
    });
}
int conv_set_option(const char* name, int value) {
    if (strcmp(name, "gv16_max_columns")) return 1;
    gv_read_env();
    g_gv_maxn.store(value < 0 ? 0 : value);
    return 0;
}
bool conv_gv16_preferred(const ConvArgs& a) {
    gv_read_env();
    if (!g_gv.load() || !conv_mfma_supported(a) || a.n_total < 1 || a.n_total > g_gv_maxn.load() || a.ktot % 16) return false;
    const long long steps = a.ktot / 16;
    const long long otiles = (long long)a.groups * ((a.cout_g + 31) / 32) * ((a.n_total + 31) / 32);
    if (steps > 15 * 24 || otiles > kGvCounters) return false;                // at most 15 slices of at most 24 steps
    const long long S = steps <= 12 * 15 ? (steps + 11) / 12 : (steps + 23) / 24;
    size_t part = 0;
    (void)conv_mfma_workspace_bytes(&part);
    return otiles * S * 4096ll <= (long long)part && otiles * S <= 65535 * 8;  // the slices' partial sums fit the workspace
}

int launch_conv_gv16(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    GvArgs gv;
    gv.mt32_per_g = (a.cout_g + 31) / 32;
    gv.tiles = a.groups * gv.mt32_per_g;
    gv.ntiles = (a.n_total + 31) / 32;
    gv.ksteps = a.ktot / 16;
    gv.ksteps_packed = (a.ktot + 63) / 64 * 4;
    gv.cpt16 = a.cin_g / 16;
    const bool small = gv.ksteps <= 12 * 15;
    gv.S = small ? (gv.ksteps + 11) / 12 : (gv.ksteps + 23) / 24;
    gv.inv_t_out = 1.0f / (float)a.t_out;
    gv.in_bytes = (unsigned)((unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull);
    gv.w_bytes = (unsigned)((unsigned long long)a.groups * gv.mt32_per_g * gv.ksteps_packed * 2048ull);
    size_t flags_offset = 0;
    const size_t need = conv_mfma_workspace_bytes(&flags_offset);
    const size_t part = (size_t)gv.tiles * gv.ntiles * gv.S * 4096;
    if (!ws.ptr || ws.bytes < need || part > flags_offset) return fail(ADK_ERR_STATE, "conv_gv16: workspace missing or too small");
    gv.ws = ws.ptr; gv.ws_bytes = (unsigned)part;
    gv.counters = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws.ptr) + need - (size_t)kGvCounters * sizeof(unsigned));
    gv.err = conv_err_word(a);
    const unsigned grid = (unsigned)(gv.tiles * gv.ntiles * gv.S);
    ConvArgs b = a;
    int act = a.act_in;
    if (a.in_sh) { b.in = a.in_sh; act = kActPre; }
#define ADK_GV_LAUNCH(ACT_) do { if (small) hipLaunchKernelGGL((conv_gv16_kernel<ACT_, 12>), dim3(grid), dim3(64), 0, s, b, gv); \
                                 else hipLaunchKernelGGL((conv_gv16_kernel<ACT_, 24>), dim3(grid), dim3(64), 0, s, b, gv); } while (0)
    if (act == kActPre) ADK_GV_LAUNCH(kActPre);
    else if (act == ADK_ACT_ELU) ADK_GV_LAUNCH(ADK_ACT_ELU);
    else if (act == ADK_ACT_LEAKY) ADK_GV_LAUNCH(ADK_ACT_LEAKY);
    else if (act == ADK_ACT_NONE) ADK_GV_LAUNCH(ADK_ACT_NONE);
    else return fail(ADK_ERR_ARG, "conv: unsupported input activation for the MFMA kernel");
#undef ADK_GV_LAUNCH
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

int launch_conv_sk16(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    if (a.n_total == 0) return ADK_OK;
    (void)conv_mfma_workspace_bytes(nullptr);
    if (conv_gv16_preferred(a)) return launch_conv_gv16(a, s, ws);
    // ADK_SK16_KD=2: 128-deep chunks.  Measured: single launches of the small layers 20-30 % faster (transposed convs
    // 26.6 -> 18.8 us), single-stream latency 1.09 -> 1.03 ms, but 67.6 KB of LDS per workgroup keeps concurrently
    // running programs off the CU: 3-stream pipeline 196 k vs 204 k frames/s.  Default 64-deep.
    static int kd = -1;
    if (kd < 0) { const char* e = getenv("ADK_SK16_KD"); kd = (e && atoi(e) == 2) ? 2 : 1; }
    const int pick = conv_sk16_pick(a);
    if (kd == 2 && a.ktot > 64) {
        // only the 64x64 tile fits 128-deep chunks without spilling (247 VGPRs; the wider tiles need > 256)
        if (pick == 2) return launch_cfg<2, 2, 1, true, 2>(a, s, ws);
    }
    switch (pick) {
        case 0: return launch_cfg<4, 1, 2, true>(a, s, ws);
        case 1: return launch_cfg<4, 1, 4, true>(a, s, ws);
        case 2: return launch_cfg<2, 2, 1, true>(a, s, ws);
        case 3: return launch_cfg<2, 2, 2, true>(a, s, ws);
        case 4: return launch_cfg<1, 4, 1, true>(a, s, ws);
        case 6: return launch_cfg<4, 2, 2, true>(a, s, ws);
        default: return launch_cfg<1, 4, 2, true>(a, s, ws);
    }
}

int launch_pack_weights(const float* w, float* out, int groups, int cout_g, int ktot, hipStream_t s) {
    const long long total = (long long)groups * ((cout_g + 31) / 32) * ((ktot + 63) / 64 * 8) * 256;
    if (total == 0) return ADK_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(pack_weights_kernel, dim3((unsigned)blocks), dim3(256), 0, s, w, out, groups, cout_g, ktot);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

}  // namespace adk
