// Implicit-GEMM causal conv on the fp32 matrix cores (v_mfma_f32_32x32x2_f32; exact f32 FMA chain).
//
//   D[m][n] = sum_k W[m][k] * X[k][n],   m = output channel (x phase for transposed convs),
//   n = (stream b, output step t),       k = (tap j, input channel ci)
//
// Replaces F.conv1d / F.conv_transpose1d as called from CausalConv1d.inference /
// CausalConvTranspose1d.inference (layers/conv_layer.py:153-156, 194-197) and Conv1d1x1 (:28-32),
// fused with the reference's surrounding element-wise ops (input ELU/LeakyReLU, bias, residual add).
//
// Many independent streams make N = B*T large even when one stream contributes a single step, so
// every conv of the path with Cin % 32 == 0 is a GEMM with M in 32..1280, K in 32..2816.
//   * 256 threads = 4 waves (64 lanes each), tile BM x BN, K-chunk = one tap x 32 channels
//   * W chunk and gathered X chunk are staged in LDS (row stride 36 floats: conflict-free
//     ds_write_b128 / ds_read_b128), double buffered, one barrier per chunk
//   * X columns are gathered straight from the channel-last state rings: 128 contiguous bytes per
//     (column, tap) -- coalesced along the channel axis; the causal history is just earlier rows
//   * the input activation is applied once per staged element, before it lands in LDS
//   * each lane reads 4 consecutive k per ds_read_b128 and feeds 4 MFMAs (the k order inside a
//     chunk is permuted identically for both operands, which only reorders the exact f32 sum)
#include "adk_common.h"
#include <cstdlib>

namespace adk {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int KC = 32;    // K chunk
constexpr int LDK = 36;   // padded LDS row stride (floats)

template <int BM, int BN, int WGM, int WGN>
__global__ __launch_bounds__(256) void conv_mfma_kernel(ConvArgs a) {
    constexpr int WM = BM / WGM, WN = BN / WGN;       // wave tile
    constexpr int MI = WM / 32, NJ = WN / 32;         // 32x32 MFMA tiles per wave
    constexpr int RA = BM / 32, RB = BN / 32;         // staging rounds (32 rows per round)
    static_assert(WGM * WGN == 4 && WM % 32 == 0 && WN % 32 == 0, "tile config");
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                         // [2][BM*LDK]
    float* Bs = smem + 2 * BM * LDK;          // [2][BN*LDK]

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / WGN, wn = wave % WGN;
    const int m_tiles = (a.cout_g + BM - 1) / BM;
    const int g = blockIdx.y / m_tiles;
    const int m0 = (blockIdx.y - g * m_tiles) * BM;
    const int n0 = blockIdx.x * BN;
    const int srow = tid >> 3, quad = tid & 7;        // staging: 8 lanes x float4 per 32-float row

    // per-thread column bookkeeping for the X gather (k-invariant)
    const float* colp[RB];
    int trow[RB];
#pragma unroll
    for (int r = 0; r < RB; ++r) {
        const int n = n0 + srow + 32 * r;
        if (n < a.n_total) {
            const int b = n / a.t_out, t = n - b * a.t_out;
            colp[r] = a.in + (size_t)b * a.in_rows * a.in_ch + a.in_choff + g * a.in_gstride + 4 * quad;
            trow[r] = a.in_row0 + t * a.stride;
        } else {
            colp[r] = nullptr;
            trow[r] = 0;
        }
    }
    const float* wp[RA];
#pragma unroll
    for (int r = 0; r < RA; ++r) {
        const int m = m0 + srow + 32 * r;
        wp[r] = (m < a.cout_g) ? a.w + (size_t)(g * a.cout_g + m) * a.ktot + 4 * quad : nullptr;
    }

    f32x16 acc[MI][NJ];
#pragma unroll
    for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NJ; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    float4 ra[RA], rb[RB];
    const int nchunks = a.ktot / KC;

    auto gload = [&](int kc) {
        const int k0 = kc * KC;
        const int tap = k0 / a.cin_g;
        const int ci0 = k0 - tap * a.cin_g;
        const int roff = tap * a.dilation;
#pragma unroll
        for (int r = 0; r < RA; ++r)
            ra[r] = wp[r] ? *reinterpret_cast<const float4*>(wp[r] + k0) : make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            int row = trow[r] + roff;
            if (row >= a.in_rows) row -= a.in_rows;
            rb[r] = colp[r] ? *reinterpret_cast<const float4*>(colp[r] + (size_t)row * a.in_ch + ci0)
                            : make_float4(0.f, 0.f, 0.f, 0.f);
        }
    };
    auto lstore = [&](int buf) {
        float* Ab = As + buf * BM * LDK;
        float* Bb = Bs + buf * BN * LDK;
#pragma unroll
        for (int r = 0; r < RA; ++r)
            *reinterpret_cast<float4*>(Ab + (srow + 32 * r) * LDK + 4 * quad) = ra[r];
#pragma unroll
        for (int r = 0; r < RB; ++r) {
            float4 v = rb[r];
            if (a.act_in != ADK_ACT_NONE) {
                v.x = act_apply(v.x, a.act_in, a.slope);
                v.y = act_apply(v.y, a.act_in, a.slope);
                v.z = act_apply(v.z, a.act_in, a.slope);
                v.w = act_apply(v.w, a.act_in, a.slope);
            }
            *reinterpret_cast<float4*>(Bb + (srow + 32 * r) * LDK + 4 * quad) = v;
        }
    };

    gload(0);
    lstore(0);
    __syncthreads();

    const int l31 = lane & 31, lh = lane >> 5;
    for (int kc = 0; kc < nchunks; ++kc) {
        const int cur = kc & 1;
        if (kc + 1 < nchunks && !(a.dbg & 1)) gload(kc + 1);          // global loads in flight under the MFMAs
        const float* Ab = As + cur * BM * LDK + (wm * WM + l31) * LDK + 4 * lh;
        const float* Bb = Bs + cur * BN * LDK + (wn * WN + l31) * LDK + 4 * lh;
        if (!(a.dbg & 2))
#pragma unroll
        for (int q = 0; q < KC / 8; ++q) {
            float4 av[MI], bv[NJ];
#pragma unroll
            for (int i = 0; i < MI; ++i) av[i] = *reinterpret_cast<const float4*>(Ab + i * 32 * LDK + q * 8);
#pragma unroll
            for (int j = 0; j < NJ; ++j) bv[j] = *reinterpret_cast<const float4*>(Bb + j * 32 * LDK + q * 8);
#pragma unroll
            for (int i = 0; i < MI; ++i)
#pragma unroll
                for (int j = 0; j < NJ; ++j) {
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].x, bv[j].x, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].y, bv[j].y, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].z, bv[j].z, acc[i][j], 0, 0, 0);
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[i].w, bv[j].w, acc[i][j], 0, 0, 0);
                }
        }
        if (!(a.dbg & 4)) {
        if (kc + 1 < nchunks) lstore(cur ^ 1);
        __syncthreads();
        }
    }

    // epilogue: lane holds column n = ..+(lane&31), rows 8*qd + 4*(lane>>5) + {0..3} of each 32x32 tile
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
        const int n = n0 + wn * WN + j * 32 + l31;
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
        for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = m0 + wm * WM + i * 32 + 8 * qd + 4 * lh;     // row within the group
                if (ml >= a.cout_g) continue;
                const int mg = g * a.cout_g + ml;
                float4 v = make_float4(acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]);
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

bool conv_mfma_supported(const ConvArgs& a) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (a.cin_g % KC != 0 || a.cout_g % 4 != 0 || a.cout_real % 4 != 0) return false;
    if (a.in_ch % 4 || a.in_choff % 4 || a.in_gstride % 4 || a.out_ch % 4 || a.out_choff % 4) return false;
    if (!al16(a.in) || !al16(a.out) || !al16(a.w) || (a.bias && !al16(a.bias))) return false;
    if (a.res && (a.res_ch % 4 || a.res_choff % 4 || a.res_gstride % 4 || !al16(a.res))) return false;
    return true;
}

namespace {
struct Cfg { int bm, bn; };
template <int BM, int BN, int WGM, int WGN>
int launch_cfg(const ConvArgs& a, hipStream_t s) {
    static bool attr_set = false;
    constexpr size_t lds = 2ull * (BM + BN) * LDK * sizeof(float);
    auto kern = conv_mfma_kernel<BM, BN, WGM, WGN>;
    if (!attr_set) {
        ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr_set = true;
    }
    const int m_tiles = (a.cout_g + BM - 1) / BM;
    dim3 grid((a.n_total + BN - 1) / BN, m_tiles * a.groups);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, s, a);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}
}  // namespace

// Tile choice: the largest tile that still yields >= 2 workgroups per CU (256 CUs); M padding
// waste is avoided by matching BM to cout_g.  ADK_CONV_CFG=<0..5> forces a config (tuning aid).
static const Cfg kCfgs[6] = {{128, 128}, {64, 128}, {32, 256}, {64, 64}, {128, 32}, {32, 128}};

static int g_forced_cfg = -2;     // -2: not initialised (read ADK_CONV_CFG), -1: heuristic

void conv_mfma_force_cfg(int cfg) { g_forced_cfg = cfg; }

int conv_mfma_pick(const ConvArgs& a) {
    if (g_forced_cfg == -2) { const char* e = getenv("ADK_CONV_CFG"); g_forced_cfg = e ? atoi(e) : -1; }
    const int forced = g_forced_cfg;
    if (forced >= 0 && forced <= 5) return forced;
    // Measured on MI355X (profiles/r1_cfg_sweep.md): with one barrier per 32-deep K chunk the 32x32
    // wave tile (one accumulator per wave, >= 3 workgroups per CU) beats the larger tiles on every
    // layer of the path, so pick among the three 4-wave arrangements of it by the M extent.
    if (a.cout_g % 128 == 0) return 4;      // 128 x 32
    if (a.cout_g % 64 == 0) return 3;       // 64 x 64
    return 5;                               // 32 x 128
}

const char* conv_mfma_cfg_name(int pick) {
    static const char* names[6] = {"conv_mfma<128,128>", "conv_mfma<64,128>", "conv_mfma<32,256>",
                                   "conv_mfma<64,64>", "conv_mfma<128,32>", "conv_mfma<32,128>"};
    return names[pick];
}

int launch_conv_mfma(const ConvArgs& a0, hipStream_t s) {
    if (a0.n_total == 0) return ADK_OK;
    ConvArgs a = a0;
    { static int dbg = -1; if (dbg < 0) { const char* e = getenv("ADK_CONV_DBG"); dbg = e ? atoi(e) : 0; } a.dbg = dbg; }
    switch (conv_mfma_pick(a)) {
        case 0: return launch_cfg<128, 128, 2, 2>(a, s);
        case 1: return launch_cfg<64, 128, 2, 2>(a, s);
        case 2: return launch_cfg<32, 256, 1, 4>(a, s);
        case 3: return launch_cfg<64, 64, 2, 2>(a, s);
        case 4: return launch_cfg<128, 32, 4, 1>(a, s);
        default: return launch_cfg<32, 128, 1, 4>(a, s);
    }
}

}  // namespace adk
