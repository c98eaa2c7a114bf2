// conv_oc16 -- the tail of the HiFi-GAN vocoder as ONE streaming kernel:
//     c = blocks[-1].conv_out(x)               MultiGroupConv1d.inference, 1x1 conv over the groups (96 -> 32)   models/vocoder/modules/multi_fusion.py:139-141
//     y = tanh(output_conv(LeakyReLU(c)))      CausalConv1d 32 -> 1, K 7, + bias                                models/vocoder/HiFiGAN.py:292-296
// Until round 6 two launches: conv_sk16<32x128> (20.7 us per 256-stream step: 29.5 MB in, 9.8 MB out) and conv_cout1 (11.8 us: the same
// 9.8 MB in again, 0.3 MB out).  Here the 32-channel tensor between them never exists in memory -- only its last 6 steps are stored, the next
// call's history -- and the launch is the kind of kernel conv_ou16 is: every byte by LDS-DMA, coalesced, hand-counted waits.
//
// One workgroup = one time slice of one stream: SL <= 122 new steps + the 6 steps in front of them (the K7 conv's receptive field), 128 columns
// of a 32 x 96 GEMM on v_mfma_f32_32x32x16_f16 (wave w = columns 32 w ..), ~46 KB of LDS: three workgroups share a CU, a 256-stream frame
// (300 steps = 3 slices of 100) is 768 workgroups = one round.  The 6 leading columns of slices 1.. are recomputed from the input rows (they
// belong to this call); those of slice 0 are the 6 steps the previous call left in front of the cursor of the 32-channel ring.
//   * W1 (12 KB of split-f16 fragments), {history of c, bias 1, the 224 weights of the output conv} and a wave's 32 rows x 384 B of
//     activations -- as three column blocks of 32 channels through two 4-KiB slots, XOR-swizzled through the DMA's source addresses;
//   * GEMM 1 leaves c in the accumulators; raw c of the call's last 6 steps goes to the ring, LeakyReLU(c) as f32 over the wave's slots
//     (row stride 144 B: conflict-free 16-byte reads along the time axis);
//   * the output conv is exact f32 on the vector ALU, one lane per step, taps outer / channels inner -- the order of conv_cout1_kernel, so
//     that part of the result is bit-identical to the two-launch form -- then + bias, tanh, one coalesced store.
// GEMM 1 sums the same products in the same 16-k chunks as conv_sk16 does for this conv (K = 96: no stream-K cut).
#include "adk_common.h"
#include <type_traits>
#include <cstdlib>

namespace adk {
namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8u __attribute__((ext_vector_type(8)));

constexpr float kOcLoScale = 2048.f, kOcLoInv = 1.f / 2048.f;
constexpr int OC_CIN = 96, OC_CM = 32, OC_TAPS = 7, OC_HALO = OC_TAPS - 1;
constexpr int OC_COLS = 128;                     // GEMM columns per workgroup (4 waves x 32)
constexpr int OC_SLMAX = OC_COLS - OC_HALO;      // new steps per workgroup at most
constexpr int OC_NCB = OC_CIN / 32;              // column blocks of 32 channels
constexpr int OC_SLOT = 4096;                    // 32 rows x 128 B
constexpr int OC_RING = 2 * OC_SLOT;             // per wave; after GEMM 1 the wave's 32 rows of act(c), 144 B apart
constexpr int OC_CROW = 4 * OC_CM + 16;          // row stride of act(c): 128 B + 16 B pad = 9 x 16 B
constexpr int OC_W1B = (OC_CIN / 16) * 2048;     // 6 chunks of 16 k: [hi 1 KiB | lo 1 KiB] each
constexpr int OC_STAGE = 2048;                   // [6 x 128 B history of c][128 B bias 1][128 B unused] [896 B output-conv weights][128 B unused]
constexpr int OC_LDS = OC_W1B + 4 * OC_RING + OC_STAGE;

struct OcArgs { int n_slices, sl; int* err; };

#define OC_DMA16(gptr, m0val) do { unsigned m0_keep_; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                                                            : "=&s"(m0_keep_) : "v"(gptr), "s"(m0val) : "memory"); } while (0)      /* M0 is the compiler's: put back */
template <int N> __device__ __forceinline__ void oc_wait_vm() {
    if constexpr (N == 0) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    else if constexpr (N == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else static_assert(N == 0, "add the count");
}

template <int ACT>
__device__ __forceinline__ float oc_act(float x, float slope) {
    if (ACT == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (ACT == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    return x;
}

template <int ACT>
__global__ __launch_bounds__(256, 3) void conv_oc16_kernel(ConvArgs a1, ConvArgs a2, OcArgs u) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x / u.n_slices, sl_i = blockIdx.x - b * u.n_slices;
    const int T = a1.t_out;
    const int t0 = sl_i * u.sl;
    const int nt = min(u.sl, T - t0);                      // new steps of this slice (>= 1: the host's slice count)
    const int ncol = nt + OC_HALO;                         // GEMM columns in use: column j <-> step t0 - 6 + j
    const bool act_w = __builtin_amdgcn_readfirstlane(wave * 32 < ncol ? 1 : 0) != 0;

    unsigned char* w1l = lds;                              // [6 chunks][hi | lo][64 lanes][16 B]
    unsigned char* ring = lds + OC_W1B + wave * OC_RING;
    unsigned char* ring_all = lds + OC_W1B;
    unsigned char* stage = lds + OC_W1B + 4 * OC_RING;
    typedef unsigned char __attribute__((address_space(3)))* lds_u8_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_u8_t)lds;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- every byte by LDS-DMA, oldest first: wave 0 {history of c | bias 1} and the output conv's weights (2), W1 (3 per wave), column
    // blocks 0 and 1 (4 each); block 2 follows when block 0 has been read.  (Wave 0's extra instructions are its oldest: one set of counts.)
    if (wave == 0) {
        const unsigned char* dummy = reinterpret_cast<const unsigned char*>(a1.wfrag) + lane16;
        const unsigned char* src = dummy;
        if (lane < 8 * OC_HALO) {                          // 6 rows x 8 pieces: what the previous call left in front of the cursor (zeros after a reset)
            int row = a2.in_row0 + (lane >> 3);
            if (row >= a2.in_rows) row -= a2.in_rows;
            src = reinterpret_cast<const unsigned char*>(a2.in + ((size_t)b * a2.in_rows + row) * a2.in_ch + a2.in_choff) + 16 * (lane & 7);
        } else if (lane < 8 * OC_HALO + 8) { if (a1.bias) src = reinterpret_cast<const unsigned char*>(a1.bias) + 16 * (lane - 8 * OC_HALO); }
        OC_DMA16(src, lds0 + (unsigned)(stage - lds));
        src = lane < OC_TAPS * OC_CM / 4 ? reinterpret_cast<const unsigned char*>(a2.w) + lane16 : dummy;
        OC_DMA16(src, lds0 + (unsigned)(stage - lds) + 1024u);
    }
    {
        const unsigned char* g1 = reinterpret_cast<const unsigned char*>(a1.wfrag) + (size_t)tid * 16;
        const unsigned l1 = lds0 + (unsigned)wave * 1024u;
#pragma unroll
        for (int i = 0; i < OC_W1B / 4096; ++i) OC_DMA16(g1 + 4096 * i, l1 + 4096u * i);
    }
    float chk = 0.f;                                        // stays 0 while every c is finite
    if (!act_w) {
        oc_wait_vm<0>();
        __syncthreads();
    } else {
        // this lane's source of piece (instruction j, column block cb): row r = 8 j + lane / 8 of the tile, 16-byte piece (lane & 7) ^ ((r >> 1) & 7)
        const unsigned char* xsrc[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int r = 8 * j + (lane >> 3);
            int tr = t0 - OC_HALO + wave * 32 + r;
            tr = max(0, min(tr, T - 1));                    // columns in front of the call (slice 0: replaced by the history) / past the slice: any valid row
            int row = a1.in_row0 + tr;
            row %= a1.in_rows;
            xsrc[j] = reinterpret_cast<const unsigned char*>(a1.in + ((size_t)b * a1.in_rows + row) * a1.in_ch + a1.in_choff) + 16 * ((lane & 7) ^ ((r >> 1) & 7));
        }
        const unsigned ring0 = lds0 + (unsigned)(ring - lds);
        auto issue_block = [&](int cb) __attribute__((always_inline)) {
            const unsigned dst = ring0 + (unsigned)(cb & 1) * OC_SLOT;
#pragma unroll
            for (int j = 0; j < 4; ++j) OC_DMA16(xsrc[j] + 128 * cb, dst + 1024u * j);
        };
        issue_block(0); issue_block(1);

        // ---- GEMM 1: c[m][t] = sum_k W1[m][k] x[k][t]: 32 rows x this wave's 32 columns, K = 96 = three column blocks of two 16-k chunks ----
        f32x16 am, ac;
#pragma unroll
        for (int e = 0; e < 16; ++e) { am[e] = 0.f; ac[e] = 0.f; }
        const unsigned swz = (unsigned)((l31 >> 1) & 7);
#pragma unroll
        for (int cb = 0; cb < OC_NCB; ++cb) {
            // in flight at most (11 issued up front, + 4 for block 2):
            if (cb == 0) { oc_wait_vm<4>(); __syncthreads(); }          // {stage, W1, block 0} landed here -- and, behind the barrier, everybody's W1 and wave 0's stage
            else if (cb == 1) oc_wait_vm<4>();                          // block 1; block 2 may stay in flight
            else oc_wait_vm<0>();
            const unsigned char* xs = ring + (cb & 1) * OC_SLOT + l31 * 128;
            float4 xr[2][2];
#pragma unroll
            for (int sc = 0; sc < 2; ++sc)
#pragma unroll
                for (int h = 0; h < 2; ++h) xr[sc][h] = *reinterpret_cast<const float4*>(xs + 16 * ((unsigned)(4 * sc + 2 * lh + h) ^ swz));
            if (cb == 0) {
                asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");     // the reads of slot 0 have returned: it may be overwritten
                issue_block(2);
            }
#pragma unroll
            for (int sc = 0; sc < 2; ++sc) {
                const unsigned char* wp = w1l + (size_t)(2 * cb + sc) * 2048 + lane * 16;
                const f16x8u Ah = *reinterpret_cast<const f16x8u*>(wp);
                const f16x8u Al = *reinterpret_cast<const f16x8u*>(wp + 1024);
                const float x[8] = {xr[sc][0].x, xr[sc][0].y, xr[sc][0].z, xr[sc][0].w, xr[sc][1].x, xr[sc][1].y, xr[sc][1].z, xr[sc][1].w};
                f16x8u bh, bl;
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const _Float16 h = (_Float16)x[e];
                    bh[e] = h; bl[e] = (_Float16)((x[e] - (float)h) * kOcLoScale);
                }
                am = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh, am, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl, ac, 0, 0, 0);
                ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh, ac, 0, 0, 0);
            }
        }
        // c (+ bias): the call's last 6 steps raw to the ring (the next call's history); act(c) as f32 over this wave's slots (every block
        // of them has been read).  Lane (column l31, half lh) holds channels 8 qd + 4 lh + {0..3}.
        const int j = wave * 32 + l31;                      // column; step t0 - 6 + j
        const int t = t0 - OC_HALO + j;
        const bool own = j >= OC_HALO && j < ncol;           // a new step of this slice (the leading 6 columns belong to the slice in front)
        const bool to_ring = own && t >= T - OC_HALO;
        const bool from_hist = t < 0;                        // slice 0 only: the previous call's steps
        const float* hist = reinterpret_cast<const float*>(stage) + (from_hist ? j : 0) * OC_CM;
        const float* b1l = reinterpret_cast<const float*>(stage + 128 * OC_HALO);
        unsigned char* crow = ring + l31 * OC_CROW;
        int orow = a1.out_cursor + (to_ring ? t : 0);
        orow %= a1.out_rows;
        float* cdst = a1.out + ((size_t)b * a1.out_rows + orow) * a1.out_ch + a1.out_choff;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int ml = 8 * qd + 4 * lh;
            float4 v = make_float4(fmaf(ac[4 * qd], kOcLoInv, am[4 * qd]), fmaf(ac[4 * qd + 1], kOcLoInv, am[4 * qd + 1]),
                                   fmaf(ac[4 * qd + 2], kOcLoInv, am[4 * qd + 2]), fmaf(ac[4 * qd + 3], kOcLoInv, am[4 * qd + 3]));
            chk = fmaf(v.x, 0.f, chk); chk = fmaf(v.y, 0.f, chk); chk = fmaf(v.z, 0.f, chk); chk = fmaf(v.w, 0.f, chk);
            if (a1.bias) {
                const float4 bb = *reinterpret_cast<const float4*>(b1l + ml);
                v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            }
            if (from_hist) v = *reinterpret_cast<const float4*>(hist + ml);
            if (to_ring) *reinterpret_cast<float4*>(cdst + ml) = v;
            v.x = oc_act<ACT>(v.x, a2.slope); v.y = oc_act<ACT>(v.y, a2.slope); v.z = oc_act<ACT>(v.z, a2.slope); v.w = oc_act<ACT>(v.w, a2.slope);
            *reinterpret_cast<float4*>(crow + 4 * ml) = v;
        }
    }
    __syncthreads();

    // ---- the output conv, exact f32: one lane per new step; y[t] = act_out(bias + sum_k sum_c w[k][c] act(c)[t - 6 + k][c]), k outer, c inner ----
    if (tid < nt) {
        const float4* wl = reinterpret_cast<const float4*>(stage + 1024);
        float acc = 0.f;
#pragma unroll
        for (int k = 0; k < OC_TAPS; ++k) {
            const int j = tid + k;
            const float4* cr = reinterpret_cast<const float4*>(ring_all + (j >> 5) * OC_RING + (j & 31) * OC_CROW);
#pragma unroll
            for (int c4 = 0; c4 < OC_CM / 4; ++c4) {
                const float4 w = wl[k * (OC_CM / 4) + c4], x = cr[c4];
                acc = fmaf(w.x, x.x, acc); acc = fmaf(w.y, x.y, acc); acc = fmaf(w.z, x.z, acc); acc = fmaf(w.w, x.w, acc);
            }
        }
        if (a2.bias) acc += a2.bias[0];
        acc = act_apply(acc, a2.act_out, 0.f);
        int orow = a2.out_cursor + t0 + tid;
        orow %= a2.out_rows;
        a2.out[((size_t)b * a2.out_rows + orow) * a2.out_ch + a2.out_choff] = acc;
    }
    if (!(chk == 0.f)) atomicOr(u.err, 8);
}

bool oc_aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace

// a1: the 1x1 conv 96 -> 32, a2: the K7 conv 32 -> 1 that reads exactly what a1 writes
bool conv_oc16_fusable(const ConvArgs& a1, const ConvArgs& a2) {
    if (!a1.wfrag || a1.taps != 1 || a1.stride != 1 || a1.up != 1 || a1.groups != 1 || a1.res || a1.act_in != ADK_ACT_NONE || a1.act_out != ADK_ACT_NONE) return false;
    if (a1.cin_g != OC_CIN || a1.cout_g != OC_CM || a1.cout_real != OC_CM) return false;
    if (!a2.w || a2.taps != OC_TAPS || a2.dilation != 1 || a2.stride != 1 || a2.up != 1 || a2.groups != 1 || a2.cin_g != OC_CM || a2.cout_g != 1 || a2.res) return false;
    if (a2.act_in == ADK_ACT_TANH) return false;
    if (a1.batch != a2.batch || a1.t_out != a2.t_out || a1.t_out < 1) return false;
    if (a2.in != a1.out || a2.in_rows != a1.out_rows || a2.in_ch != a1.out_ch || a2.in_choff != a1.out_choff || a1.out_rows < a1.t_out + OC_HALO) return false;
    if (a2.in_row0 != (a1.out_cursor + a1.out_rows - OC_HALO) % a1.out_rows) return false;       // six steps of history, right in front of the new rows
    if ((a1.in_ch % 4) || (a1.in_choff % 4) || (a1.out_ch % 4) || (a1.out_choff % 4) || !oc_aligned16(a1.in) || !oc_aligned16(a1.out) || !oc_aligned16(a1.wfrag) ||
        !oc_aligned16(a2.w)) return false;
    if (a1.bias && !oc_aligned16(a1.bias)) return false;
    return true;
}

int launch_conv_oc16(const ConvArgs& a1, const ConvArgs& a2, hipStream_t s) {
    if (!conv_oc16_fusable(a1, a2)) return ADK_ERR_STATE;
    if (a1.n_total == 0) return ADK_OK;
    OcArgs u;
    u.n_slices = (a1.t_out + OC_SLMAX - 1) / OC_SLMAX;
    u.sl = (a1.t_out + u.n_slices - 1) / u.n_slices;
    u.n_slices = (a1.t_out + u.sl - 1) / u.sl;            // (no empty slice)
    u.err = conv_err_word(a1);
    const long long blocks = (long long)a1.batch * u.n_slices;
    if (blocks > 0x7fffffffLL) return ADK_ERR_STATE;
    auto go = [&](auto act) -> int {
        constexpr int ACT = decltype(act)::value;
        hipLaunchKernelGGL(conv_oc16_kernel<ACT>, dim3((unsigned)blocks), dim3(256), OC_LDS, s, a1, a2, u);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    };
    if (a2.act_in == ADK_ACT_ELU) return go(std::integral_constant<int, ADK_ACT_ELU>());
    if (a2.act_in == ADK_ACT_LEAKY) return go(std::integral_constant<int, ADK_ACT_LEAKY>());
    return go(std::integral_constant<int, ADK_ACT_NONE>());
}

}  // namespace adk
