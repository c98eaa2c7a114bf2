// Residual-VQ encode / lookup and the ring writer.
//   encode: ResidualVQ.forward_index over VectorQuantize.forward_index (layers/vq_module.py:136-149, 90-104)
//   lookup: ResidualVQ.lookup (layers/vq_module.py:159-161)
//   ring_write: torch.cat of new rows onto the state (layers/conv_layer.py:154) + HiFiGAN decode_norm
//               (models/vocoder/HiFiGAN.py:276-279)
#include <cstring>
#include "adk_common.h"
#include <cstdlib>
#include <mutex>
#include <atomic>
#include <type_traits>
#include <vector>

namespace adk {

constexpr int RVQ_DIM_MAX = 128;

__device__ int g_adk_flags = 0;      // bit 0: rvq_lookup saw an out-of-range index; bit 1: stream-K publish flag timeout

static int* g_flag_ptr[kMaxDevices] = {};   // one word per device: the symbol's address differs from device to device

int* flags_word() {
    const int d = current_device();
    if (!g_flag_ptr[d]) (void)hipGetSymbolAddress(reinterpret_cast<void**>(&g_flag_ptr[d]), HIP_SYMBOL(g_adk_flags));
    return g_flag_ptr[d];
}

// (value, index) arg-max with "greater value, else smaller index" -- matches `(-dist).max(1)` on the
// reference's CPU path, which returns the lowest index among equal maxima.
__device__ __forceinline__ void argmax_merge(float& v, int& i, float v2, int i2) {
    if (v2 > v || (v2 == v && i2 < i)) { v = v2; i = i2; }
}

// One workgroup = RB rows (b,t) x all codes: 1024 threads, one code per thread per 1024-code slab, so
// the codebook (dim-major, codes contiguous: the reference's `embed` layout) is read with coalesced
// 256-byte wave loads and every loaded value is reused for RB rows; 16 waves per workgroup keep the
// L2 latency of the 64 dependent-free loads per stage hidden.  The 8 stages stay inside one launch
// because stage i+1 needs r - q'_i.  Wave-level (value,index) reduction by shuffles, then LDS
// across the 16 waves.
constexpr int RVQ_THREADS = 1024;
constexpr int RVQ_WAVES = RVQ_THREADS / 64;

template <int RB>
__global__ __launch_bounds__(RVQ_THREADS) void rvq_encode_kernel(const float* __restrict__ z, const float* __restrict__ embed,
                                                                 const float* __restrict__ enorm, long long* __restrict__ idx,
                                                                 float* __restrict__ zq, int n_rows, int n_q, int dim, int size) {
    __shared__ float r_sh[RB][RVQ_DIM_MAX];
    __shared__ __attribute__((aligned(16))) float r2t_sh[RVQ_DIM_MAX][RB];   // 2*r, dim-major: one 16-byte broadcast read per d for all RB rows
    __shared__ float q_sh[RB][RVQ_DIM_MAX];
    __shared__ float rn_sh[RB];
    __shared__ float red_v[RB][RVQ_WAVES];
    __shared__ int red_i[RB][RVQ_WAVES];
    __shared__ int best_sh[RB];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * RB;
    for (int e = tid; e < RB * dim; e += RVQ_THREADS) {
        const int rr = e / dim, d = e - rr * dim;
        const float v0 = (row0 + rr < n_rows) ? z[(size_t)(row0 + rr) * dim + d] : 0.f;
        r_sh[rr][d] = v0;
        r2t_sh[d][rr] = 2.f * v0;
        q_sh[rr][d] = 0.f;
    }
    __syncthreads();
    for (int st = 0; st < n_q; ++st) {
        const float* E = embed + (size_t)st * dim * size;
        const float* EN = enorm + (size_t)st * size;
        if (wave < RB) {                                  // flatten.pow(2).sum(1)   (vq_module.py:94)
            // one wave per row: squares summed by a fixed butterfly tree (the reference's own order is a
            // vectorised tree inside torch.sum; the value only shifts all distances of a row together)
            float v = (lane < dim) ? __fmul_rn(r_sh[wave][lane], r_sh[wave][lane]) : 0.f;
            if (lane + 64 < dim) v = __fadd_rn(v, __fmul_rn(r_sh[wave][lane + 64], r_sh[wave][lane + 64]));
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
            if (lane == 0) rn_sh[wave] = v;
        }
        __syncthreads();
        float bv[RB]; int bi[RB];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) { bv[rr] = -INFINITY; bi[rr] = 0x7fffffff; }
        for (int c = tid; c < size; c += RVQ_THREADS) {
            float acc[RB];
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) acc[rr] = 0.f;
            const float* Ec = E + c;
            // (2*flatten) @ embed, d ascending (vq_module.py:95); 16 independent loads in flight per batch
            // (deeper batches spill at the 1024-thread register budget and run slower)
            for (int d0 = 0; d0 < dim; d0 += 16) {
                float e[16];
#pragma unroll
                for (int u = 0; u < 16; ++u) e[u] = (d0 + u < dim) ? Ec[(size_t)(d0 + u) * size] : 0.f;
#pragma unroll
                for (int u = 0; u < 16; ++u)
                    if (d0 + u < dim) {
                        // the LDS read is a wave-wide broadcast; one b128 per d instead of RB b32 reads keeps the LDS pipe
                        // (16 waves x 64 d x RB rows per stage) off the critical path
                        if constexpr (RB == 4) {
                            const float4 r2 = *reinterpret_cast<const float4*>(&r2t_sh[d0 + u][0]);
                            acc[0] = fmaf(r2.x, e[u], acc[0]); acc[1] = fmaf(r2.y, e[u], acc[1]);
                            acc[2] = fmaf(r2.z, e[u], acc[2]); acc[3] = fmaf(r2.w, e[u], acc[3]);
                        } else {
#pragma unroll
                            for (int rr = 0; rr < RB; ++rr) acc[rr] = fmaf(r2t_sh[d0 + u][rr], e[u], acc[rr]);
                        }
                    }
            }
            const float en = EN[c];
#pragma unroll
            for (int rr = 0; rr < RB; ++rr) {             // dist = (|r|^2 - 2rE) + |E|^2 ; argmax(-dist)
                const float dist = __fadd_rn(__fsub_rn(rn_sh[rr], acc[rr]), en);
                argmax_merge(bv[rr], bi[rr], -dist, c);
            }
        }
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            float v = bv[rr]; int i = bi[rr];
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float v2 = __shfl_xor(v, off, 64);
                const int i2 = __shfl_xor(i, off, 64);
                argmax_merge(v, i, v2, i2);
            }
            if (lane == 0) { red_v[rr][wave] = v; red_i[rr][wave] = i; }
        }
        __syncthreads();
        if (tid < RB) {
            float v = red_v[tid][0]; int i = red_i[tid][0];
            for (int w = 1; w < RVQ_WAVES; ++w) argmax_merge(v, i, red_v[tid][w], red_i[tid][w]);
            best_sh[tid] = i;
            if (row0 + tid < n_rows) idx[(size_t)st * n_rows + row0 + tid] = (long long)i + (long long)size * st;
        }
        __syncthreads();
        for (int e = tid; e < RB * dim; e += RVQ_THREADS) {   // straight-through + residual (vq_module.py:101-102,143-144)
            const int rr = e / dim, d = e - rr * dim;
            const float r = r_sh[rr][d];
            const float q = E[(size_t)d * size + best_sh[rr]];
            const float qp = __fadd_rn(r, __fsub_rn(q, r));
            const float rn = __fsub_rn(r, qp);
            r_sh[rr][d] = rn;
            r2t_sh[d][rr] = 2.f * rn;
            q_sh[rr][d] = __fadd_rn(q_sh[rr][d], qp);
        }
        __syncthreads();
    }
    if (zq)
        for (int e = tid; e < RB * dim; e += RVQ_THREADS) {
            const int rr = e / dim, d = e - rr * dim;
            if (row0 + rr < n_rows) zq[(size_t)(row0 + rr) * dim + d] = q_sh[rr][d];
        }
}

// ---- v2: the same arithmetic, shorter dependency chain per stage (dim == 64, size == 1024) ----
// One thread = one code, its 64 components live in REGISTERS for the stage:
//   * all 64 component loads of a stage are in flight at once (v1: four dependent batches of 16), and the NEXT stage's are
//     issued as soon as the winner has handed its code over -- they land during the residual update and two barriers;
//   * the winning thread writes q from its registers to LDS (v1: a dependent gather from global memory after the argmax);
//   * every thread folds the 16 per-wave candidates itself (v1: RB threads + a barrier), wave rr owns row rr: it keeps r and
//     the running sum of q' in registers and forms |r|^2 for the next stage right after the update (v1: its own phase).
// Three barriers per stage instead of five, ~3 us per stage instead of ~11.  Per element the operations and their order are
// v1's (ascending-d fmaf chain, the same butterfly for |r|^2, the same (value, index) merges), so the indices are the same.
template <int RB>
__global__ __launch_bounds__(RVQ_THREADS) void rvq_encode_v2_kernel(const float* __restrict__ z, const float* __restrict__ embed,
                                                                    const float* __restrict__ enorm, long long* __restrict__ idx,
                                                                    float* __restrict__ zq, int n_rows, int n_q) {
    constexpr int D = 64, SIZE = 1024;
    __shared__ __attribute__((aligned(16))) float r2t_sh[D][RB];   // 2*r, dim-major: one broadcast read per d for all RB rows
    __shared__ __attribute__((aligned(16))) float q_sh[RB][D];
    __shared__ float rn_sh[RB];
    __shared__ float red_v[RB][RVQ_WAVES];
    __shared__ int red_i[RB][RVQ_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int row0 = blockIdx.x * RB;
    const bool owner = wave < RB;                          // wave rr owns row rr, lane = dimension
    float r_reg = 0.f, qsum = 0.f;
    float e[D];
    auto load_codes = [&](int st) {
        const float* Ec = embed + (size_t)st * D * SIZE + tid;
#pragma unroll
        for (int d = 0; d < D; ++d) e[d] = Ec[(size_t)d * SIZE];
    };
    load_codes(0);
    if (owner) {
        r_reg = (row0 + wave < n_rows) ? z[(size_t)(row0 + wave) * D + lane] : 0.f;
        r2t_sh[lane][wave] = 2.f * r_reg;
        float v = __fmul_rn(r_reg, r_reg);                 // flatten.pow(2).sum(1): the butterfly of v1
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
        if (lane == 0) rn_sh[wave] = v;
    }
    __syncthreads();
    for (int st = 0; st < n_q; ++st) {
        float acc[RB];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) acc[rr] = 0.f;
#pragma unroll
        for (int d = 0; d < D; ++d) {                      // (2*flatten) @ embed, d ascending (vq_module.py:95)
            if constexpr (RB == 4) {
                const float4 r2 = *reinterpret_cast<const float4*>(&r2t_sh[d][0]);
                acc[0] = fmaf(r2.x, e[d], acc[0]); acc[1] = fmaf(r2.y, e[d], acc[1]);
                acc[2] = fmaf(r2.z, e[d], acc[2]); acc[3] = fmaf(r2.w, e[d], acc[3]);
            } else if constexpr (RB == 2) {
                const float2 r2 = *reinterpret_cast<const float2*>(&r2t_sh[d][0]);
                acc[0] = fmaf(r2.x, e[d], acc[0]); acc[1] = fmaf(r2.y, e[d], acc[1]);
            } else {
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) acc[rr] = fmaf(r2t_sh[d][rr], e[d], acc[rr]);
            }
        }
        const float en = enorm[(size_t)st * SIZE + tid];
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {                  // dist = (|r|^2 - 2rE) + |E|^2 ; argmax(-dist), lowest index on ties
            float v = -__fadd_rn(__fsub_rn(rn_sh[rr], acc[rr]), en);
            int i = tid;
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) {
                const float v2 = __shfl_xor(v, off, 64);
                const int i2 = __shfl_xor(i, off, 64);
                argmax_merge(v, i, v2, i2);
            }
            if (lane == 0) { red_v[rr][wave] = v; red_i[rr][wave] = i; }
        }
        __syncthreads();                                   // A: candidates of all waves visible
#pragma unroll
        for (int rr = 0; rr < RB; ++rr) {
            float v = red_v[rr][0]; int i = red_i[rr][0];
#pragma unroll
            for (int w = 1; w < RVQ_WAVES; ++w) argmax_merge(v, i, red_v[rr][w], red_i[rr][w]);
            if (i == tid) {                                // the winner hands its code over from registers
#pragma unroll
                for (int d4 = 0; d4 < D / 4; ++d4)
                    *reinterpret_cast<float4*>(&q_sh[rr][4 * d4]) = make_float4(e[4 * d4], e[4 * d4 + 1], e[4 * d4 + 2], e[4 * d4 + 3]);
                if (row0 + rr < n_rows) idx[(size_t)st * n_rows + row0 + rr] = (long long)i + (long long)SIZE * st;
            }
        }
        if (st + 1 < n_q) load_codes(st + 1);              // in flight during the update below
        __syncthreads();                                   // B: q visible
        if (owner) {                                       // straight-through + residual (vq_module.py:101-102,143-144)
            const float q = q_sh[wave][lane];
            const float qp = __fadd_rn(r_reg, __fsub_rn(q, r_reg));
            r_reg = __fsub_rn(r_reg, qp);
            qsum = __fadd_rn(qsum, qp);
            r2t_sh[lane][wave] = 2.f * r_reg;
            float v = __fmul_rn(r_reg, r_reg);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
            if (lane == 0) rn_sh[wave] = v;
        }
        __syncthreads();                                   // C: residual of the next stage visible
    }
    if (zq && owner && row0 + wave < n_rows) zq[(size_t)(row0 + wave) * D + lane] = qsum;
}

// max over the lanes of a wave by DPP (no LDS traffic): after the four row steps every lane holds its 16-lane row's maximum, after the
// two broadcast steps lane 63 holds the wave's.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ float dpp_max_step(float m) {
    const int b = __float_as_int(m);
    return fmaxf(m, __int_as_float(__builtin_amdgcn_update_dpp(b, b, CTRL, ROW_MASK, 0xF, false)));
}
__device__ __forceinline__ float row_max16(float m) {
    m = dpp_max_step<0xB1, 0xF>(m);                        // quad_perm [1,0,3,2]
    m = dpp_max_step<0x4E, 0xF>(m);                        // quad_perm [2,3,0,1]
    m = dpp_max_step<0x141, 0xF>(m);                       // row_half_mirror
    return dpp_max_step<0x140, 0xF>(m);                    // row_mirror
}
// (value, lane) of the greatest value of a wave, lowest lane among equals -- what the shuffle tree of argmax_merge returns for i = lane
__device__ __forceinline__ int wave_argmax_lane(float v, float& vmax) {
    float m = row_max16(v);
    m = dpp_max_step<0x142, 0xA>(m);                       // row_bcast:15 into rows 1 and 3
    m = dpp_max_step<0x143, 0xC>(m);                       // row_bcast:31 into rows 2 and 3
    vmax = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(m), 63));
    const unsigned long long hit = __ballot(v == vmax);
    return hit ? __builtin_ctzll(hit) : 0;                 // (no lane compares equal only if every value is NaN)
}

// ---- v3: v2 with a shorter serial chain per stage (dim == 64, size == 1024, one row per workgroup) ----
// v2 keeps the 64 code registers until the winner has handed its code over through LDS (every thread folds the 16 candidates, one
// writes 256 bytes, barrier, the owner reads them).  Here only the owner wave folds the candidates and fetches the winning code from
// memory (one 64-lane gather of L2-hot lines), so the registers are free as soon as barrier A has passed: all waves request the next
// stage's 256 KB right there, and they land under the owner's residual update and barrier B.  Two barriers per stage instead of
// three; per element the same operations in the same order as v1 / v2 (the indices and the sum of codes are bit-identical).
__global__ __launch_bounds__(RVQ_THREADS) void rvq_encode_v3_kernel(const float* __restrict__ z, const float* __restrict__ embed,
                                                                    const float* __restrict__ enorm, long long* __restrict__ idx,
                                                                    float* __restrict__ zq, int n_rows, int n_q) {
    constexpr int D = 64, SIZE = 1024;
    __shared__ __attribute__((aligned(16))) float r2_sh[D];        // 2*r: read as 16 broadcast float4
    __shared__ float rn_sh;
    __shared__ float red_v[RVQ_WAVES];
    __shared__ int red_i[RVQ_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    // Round 5, few rows: blocks behind the rows are HELPERS (adk_rvq_encode launches 8 * n_help of them behind the rows padded to a multiple of 8).  A row's
    // workgroup pulls the 2 MB of codes of the 8 stages through ONE CU, from the Infinity Cache (between two frames they have left the L2s): 29 us for a
    // single row, whatever it computes.  A helper touches one dword of every 128-byte line of the codes and exits: the lines are then in the L2 of ITS XCD,
    // which -- block b runs on XCD b % 8, an observation, not a contract: a wrong guess costs the speed-up only -- is an XCD that hosts a row.
    if ((int)blockIdx.x >= n_rows) {
        const int pad = (n_rows + 7) & ~7;
        const int h = (int)blockIdx.x - pad;
        if (h < 0 || (h & 7) >= n_rows) return;              // padding block / an XCD without a row
        const int n_help = ((int)gridDim.x - pad) >> 3, sub = h >> 3;
        const unsigned char* base = reinterpret_cast<const unsigned char*>(embed);
        const int nlines = n_q * D * SIZE * 4 / 128;
        for (int l0 = sub * RVQ_THREADS + tid; l0 < nlines; l0 += 4 * n_help * RVQ_THREADS) {
            unsigned t0, t1, t2, t3;                           // four lines in flight per thread; the results only have to stay put until they have landed
            const int st = n_help * RVQ_THREADS;
            const unsigned char* p0 = base + (size_t)l0 * 128u;
            const unsigned char* p1 = base + (size_t)min(l0 + st, nlines - 1) * 128u;
            const unsigned char* p2 = base + (size_t)min(l0 + 2 * st, nlines - 1) * 128u;
            const unsigned char* p3 = base + (size_t)min(l0 + 3 * st, nlines - 1) * 128u;
            asm volatile("global_load_dword %0, %4, off\n\tglobal_load_dword %1, %5, off\n\tglobal_load_dword %2, %6, off\n\tglobal_load_dword %3, %7, off\n\ts_waitcnt vmcnt(0)"
                         : "=&v"(t0), "=&v"(t1), "=&v"(t2), "=&v"(t3) : "v"(p0), "v"(p1), "v"(p2), "v"(p3) : "memory");
        }
        return;
    }
    const int row = blockIdx.x;
    float r_reg = 0.f, qsum = 0.f;
    float e[D];
    {
        const float* Ec = embed + tid;
#pragma unroll
        for (int d = 0; d < D; ++d) e[d] = Ec[(size_t)d * SIZE];
    }
    float en = enorm[tid];
    if (wave == 0) {
        r_reg = z[(size_t)row * D + lane];
        r2_sh[lane] = 2.f * r_reg;
        float v = __fmul_rn(r_reg, r_reg);                 // flatten.pow(2).sum(1): the butterfly of v1
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
        if (lane == 0) rn_sh = v;
    }
    __syncthreads();
    for (int st = 0; st < n_q; ++st) {
        const bool has_next = st + 1 < n_q;
        const float* En = embed + (size_t)(has_next ? st + 1 : st) * D * SIZE + tid;     // (last stage: re-reads its own codes, unused)
        float acc = 0.f;
#pragma unroll
        for (int d4 = 0; d4 < D / 4; ++d4) {               // (2*flatten) @ embed, d ascending (vq_module.py:95)
            const float4 r2 = *reinterpret_cast<const float4*>(&r2_sh[4 * d4]);
            acc = fmaf(r2.x, e[4 * d4], acc); acc = fmaf(r2.y, e[4 * d4 + 1], acc);
            acc = fmaf(r2.z, e[4 * d4 + 2], acc); acc = fmaf(r2.w, e[4 * d4 + 3], acc);
        }
        float v = -__fadd_rn(__fsub_rn(rn_sh, acc), en);   // dist = (|r|^2 - 2rE) + |E|^2 ; argmax(-dist), lowest index on ties
        float vmax;
        const int best_lane = wave_argmax_lane(v, vmax);
        if (lane == 0) { red_v[wave] = vmax; red_i[wave] = (wave << 6) + best_lane; }
        __syncthreads();                                   // A: candidates of all waves visible
        float q = 0.f;
        if (wave == 0) {
            // the 16 candidates are in index order: the first wave that holds the maximum wins (greater value, else smaller index)
            const float cv = red_v[lane & 15];
            const float cmax = row_max16(cv);
            const unsigned long long hit = __ballot(cv == cmax);
            const int bi = red_i[hit ? __builtin_ctzll(hit) : 0];
            if (lane == 0) idx[(size_t)st * n_rows + row] = (long long)bi + (long long)SIZE * st;
            q = embed[((size_t)st * D + lane) * SIZE + bi];         // the winning code: one 64-lane gather of L2-hot lines, lane = dimension
        }
        // every wave's codes of the next stage: requested here, they land under the owner's update, barrier B and -- the tail of
        // them -- the first FMAs of the next stage.  In the owner wave they go out BEHIND the gather: loads of a wave return in
        // order, issued earlier (right after the FMA that frees a register) they hold the gather back for the whole stream --
        // measured 41 us per launch instead of 36; fetching the winner through the scalar cache instead (64 s_load_dword): 44 us.
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < D; ++d) e[d] = En[(size_t)d * SIZE];
        en = enorm[(size_t)(has_next ? st + 1 : st) * SIZE + tid];
        __builtin_amdgcn_sched_barrier(0);
        if (wave == 0) {                                   // straight-through + residual (vq_module.py:101-102,143-144)
            const float qp = __fadd_rn(r_reg, __fsub_rn(q, r_reg));
            r_reg = __fsub_rn(r_reg, qp);
            qsum = __fadd_rn(qsum, qp);
            r2_sh[lane] = 2.f * r_reg;
            float s2 = __fmul_rn(r_reg, r_reg);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s2 = __fadd_rn(s2, __shfl_xor(s2, off, 64));
            if (lane == 0) rn_sh = s2;
        }
        __syncthreads();                                   // B: residual of the next stage visible
    }
    if (zq && wave == 0) zq[(size_t)row * D + lane] = qsum;
}

// ---- v4: v3 with R rows per workgroup sharing the code registers (round 4) ----
// At 256 rows every CU pulled each stage's 256 KB of codes for ONE row: 64 MB per stage through the L2s, 512 MB per step -- the stage
// time of v3 (5.3 us) is that delivery (profiles/r3_load_rate.md), and in the three-stream pipeline it is bandwidth the convs need.
// Here a workgroup keeps R rows: the 64 code registers of a thread serve R dot products (R independent FMA chains, same per-row
// operations in the same order: indices and zq bit-identical), the R arg-max reductions share the two barriers of a stage (wave r
// folds the candidates of row r, fetches its winner, updates its residual), and a launch takes 256 / R workgroups.  Round 3 measured
// the naive form (rows one after the other, barriers per row) at +1.9 us per extra row and stage; batched, an extra row costs the 64
// FMAs and one DPP reduction per wave.
// rows per workgroup of the v4 kernel (ADK_RVQ_ROWS / adk_set_option("rvq_rows"): 0 = v3 at every row count, 2, 4, 1 (default) = 2 up
// to 512 rows and 4 above: one dispatch round of 256 workgroups each) and the row count from which it takes over (ADK_RVQ_V4_MIN /
// "rvq_v4_min": default 192 -- below, a workgroup per row is a partial round and faster: 26-28 us against 31.6).
// Measured alone on the chip (tools/rvq_time.py, us per launch of 8 stages; rows/wg 1 = v3 to 256 rows, the first-round kernel above):
//   rows    1/wg   2/wg   4/wg        in the three-stream pipeline at 256 rows, 2/wg against 1/wg: 282.5 k against 275.7 k frames/s
//    256    31.1   31.6   42.9        (same box, alternating): half the workgroups, half the code bytes pulled through the L2s
//    512    78.3   36.5   42.9
//   1024    79.3   66.3   46.7
// (atomics: adk_set_option may be called from another host thread than the one that launches; the environment is read exactly once)
static std::atomic<int> g_rvq_rows{1}, g_rvq_v4_min{192};
static std::once_flag g_rvq_env_once;
static void rvq_read_env() {
    std::call_once(g_rvq_env_once, [] {
        const char* e = getenv("ADK_RVQ_ROWS");
        const int v = e ? atoi(e) : 1;
        g_rvq_rows = (v == 0 || v == 2 || v == 4) ? v : 1;
        e = getenv("ADK_RVQ_V4_MIN");
        if (e) { const int m = atoi(e); g_rvq_v4_min = m < 1 ? 1 : m; }        // clamped as rvq_set_option does: 0 rows never reach the v4 grid
    });
}
int rvq_set_option(const char* name, int value) {
    rvq_read_env();
    if (!strcmp(name, "rvq_rows")) { if (value != 0 && value != 1 && value != 2 && value != 4) return -1; g_rvq_rows = value; return 0; }
    if (!strcmp(name, "rvq_v4_min")) { g_rvq_v4_min = value < 1 ? 1 : value; return 0; }
    return 1;
}

typedef float f2 __attribute__((ext_vector_type(2)));
// acc.{x,y} = fma(a.{x,y}, e.x, acc.{x,y}) / ... e.y ...: v_pk_fma_f32 with one half of the register pair e as the operand of both
// lanes (two IEEE FMAs; the compiler of this toolchain only ever selects the low half, i.e. spends a pair per code register).
__device__ __forceinline__ void pk_fma_lo(f2& acc, f2 a, f2 e) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[1,0,1]" : "+v"(acc) : "v"(a), "v"(e));
}
__device__ __forceinline__ void pk_fma_hi(f2& acc, f2 a, f2 e) {
    asm("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[0,1,0] op_sel_hi:[1,1,1]" : "+v"(acc) : "v"(a), "v"(e));
}
template <int R>
__global__ __launch_bounds__(RVQ_THREADS) void rvq_encode_v4_kernel(const float* __restrict__ z, const float* __restrict__ embed,
                                                                    const float* __restrict__ enorm, long long* __restrict__ idx,
                                                                    float* __restrict__ zq, int n_rows, int n_q) {
    constexpr int D = 64, SIZE = 1024;
    __shared__ __attribute__((aligned(16))) float r2_sh[R / 2][2 * D];   // 2*r, rows of a pair interleaved per dimension: broadcast float4 reads
    __shared__ float rn_sh[R];
    __shared__ float red_v[R][RVQ_WAVES];
    __shared__ int red_i[R][RVQ_WAVES];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int row0 = blockIdx.x * R;
    const bool owner = wave < R;                           // wave r owns row row0 + r (a row past the end: computed on row n_rows - 1, never stored)
    const int my_row = min(row0 + wave, n_rows - 1);
    float r_reg = 0.f, qsum = 0.f;
    f2 e[D / 2];                                           // code registers as pairs: the packed FMA selects a half as its broadcast operand
    {
        const float* Ec = embed + tid;
#pragma unroll
        for (int d = 0; d < D; ++d) e[d >> 1][d & 1] = Ec[(size_t)d * SIZE];
    }
    float en = enorm[tid];
    if (owner) {
        r_reg = z[(size_t)my_row * D + lane];
        r2_sh[wave >> 1][2 * lane + (wave & 1)] = 2.f * r_reg;
        float v = __fmul_rn(r_reg, r_reg);                 // flatten.pow(2).sum(1): the butterfly of v1
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) v = __fadd_rn(v, __shfl_xor(v, off, 64));
        if (lane == 0) rn_sh[wave] = v;
    }
    __syncthreads();
    for (int st = 0; st < n_q; ++st) {
        const bool has_next = st + 1 < n_q;
        const float* En = embed + (size_t)(has_next ? st + 1 : st) * D * SIZE + tid;
        f2 acc2[R / 2];
#pragma unroll
        for (int p = 0; p < R / 2; ++p) acc2[p] = (f2){0.f, 0.f};
#pragma unroll
        for (int d4 = 0; d4 < D / 4; ++d4) {               // (2*flatten) @ embed, d ascending (vq_module.py:95): a row pair per packed FMA
#pragma unroll
            for (int p = 0; p < R / 2; ++p) {              // (v_pk_fma_f32: two IEEE FMAs, each row's chain is the chain of v3)
                const float4 a = *reinterpret_cast<const float4*>(&r2_sh[p][8 * d4]);       // {row 2p, row 2p+1} of d, d+1
                const float4 b = *reinterpret_cast<const float4*>(&r2_sh[p][8 * d4 + 4]);   // ... of d+2, d+3
                pk_fma_lo(acc2[p], (f2){a.x, a.y}, e[2 * d4]);
                pk_fma_hi(acc2[p], (f2){a.z, a.w}, e[2 * d4]);
                pk_fma_lo(acc2[p], (f2){b.x, b.y}, e[2 * d4 + 1]);
                pk_fma_hi(acc2[p], (f2){b.z, b.w}, e[2 * d4 + 1]);
            }
        }
#pragma unroll
        for (int r = 0; r < R; ++r) {
            const float a = (r & 1) ? acc2[r / 2].y : acc2[r / 2].x;
            const float v = -__fadd_rn(__fsub_rn(rn_sh[r], a), en);   // dist = (|r|^2 - 2rE) + |E|^2 ; argmax(-dist), lowest index on ties
            float vmax;
            const int best_lane = wave_argmax_lane(v, vmax);
            if (lane == 0) { red_v[r][wave] = vmax; red_i[r][wave] = (wave << 6) + best_lane; }
        }
        __syncthreads();                                   // A: candidates of all waves, all rows
        float q = 0.f;
        if (owner) {
            const float cv = red_v[wave][lane & 15];
            const float cmax = row_max16(cv);
            const unsigned long long hit = __ballot(cv == cmax);
            const int bi = red_i[wave][hit ? __builtin_ctzll(hit) : 0];
            if (lane == 0 && row0 + wave < n_rows) idx[(size_t)st * n_rows + row0 + wave] = (long long)bi + (long long)SIZE * st;
            q = embed[((size_t)st * D + lane) * SIZE + bi];         // the winning code of this wave's row: one 64-lane gather, lane = dimension
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int d = 0; d < D; ++d) e[d >> 1][d & 1] = En[(size_t)d * SIZE];   // next stage's codes: land under the owners' updates and barrier B
        en = enorm[(size_t)(has_next ? st + 1 : st) * SIZE + tid];
        __builtin_amdgcn_sched_barrier(0);
        if (owner) {                                       // straight-through + residual (vq_module.py:101-102,143-144)
            const float qp = __fadd_rn(r_reg, __fsub_rn(q, r_reg));
            r_reg = __fsub_rn(r_reg, qp);
            qsum = __fadd_rn(qsum, qp);
            r2_sh[wave >> 1][2 * lane + (wave & 1)] = 2.f * r_reg;
            float s2 = __fmul_rn(r_reg, r_reg);
#pragma unroll
            for (int off = 32; off >= 1; off >>= 1) s2 = __fadd_rn(s2, __shfl_xor(s2, off, 64));
            if (lane == 0) rn_sh[wave] = s2;
        }
        __syncthreads();                                   // B: residuals of the next stage visible
    }
    if (zq && owner && row0 + wave < n_rows) zq[(size_t)(row0 + wave) * D + lane] = qsum;
}

__global__ __launch_bounds__(256) void rvq_lookup_kernel(const long long* __restrict__ idx, const float* __restrict__ codebook,
                                                         float* __restrict__ zq, int n_rows, int n_q, int dim, int n_codes) {
    const int d4 = dim / 4;
    const long long total = (long long)n_rows * d4;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(gid / d4), c = (int)(gid - (long long)row * d4);
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < n_q; ++q) {                    // torch.sum(F.embedding(idx, codebook), dim=0)
            long long i = idx[(size_t)q * n_rows + row];
            if (i < 0 || i >= n_codes) { atomicOr(&g_adk_flags, 1); i = 0; }
            const float4 e = *reinterpret_cast<const float4*>(codebook + (size_t)i * dim + 4 * c);
            s.x = __fadd_rn(s.x, e.x); s.y = __fadd_rn(s.y, e.y); s.z = __fadd_rn(s.z, e.z); s.w = __fadd_rn(s.w, e.w);
        }
        *reinterpret_cast<float4*>(zq + (size_t)row * dim + 4 * c) = s;
    }
}

__global__ __launch_bounds__(256) void ring_write_kernel(const float* __restrict__ src, float* __restrict__ ring, int rows, int channels,
                                                         int cursor, int ch_off, int src_ch, const float* __restrict__ mean,
                                                         const float* __restrict__ scale, int batch, int t) {
    const long long total = (long long)batch * t * src_ch;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(gid % src_ch);
        const long long bt = gid / src_ch;
        const int tt = (int)(bt % t), b = (int)(bt / t);
        float v = src[gid];
        if (mean) v = __fdiv_rn(__fsub_rn(v, mean[c]), scale[c]);      // (c - mean) / scale, true division
        int row = cursor + tt;
        if (row >= rows) row -= rows;
        ring[((size_t)b * rows + row) * channels + ch_off + c] = v;
    }
}

// out[b][t][c] = (((s0 + s1) + s2) ...) / n  -- the reference accumulates cs = 0.0; cs += y_i; c = cs / n
__global__ __launch_bounds__(256) void ring_mean_kernel(RingMeanArgs m) {
    const long long total = (long long)m.batch * m.t * m.channels;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(gid % m.channels);
        const long long bt = gid / m.channels;
        const int tt = (int)(bt % m.t), b = (int)(bt / m.t);
        float acc = 0.f;
        for (int i = 0; i < m.n; ++i) {
            int row = m.src_cursor[i] + tt;
            if (row >= m.src_rows[i]) row -= m.src_rows[i];
            acc = __fadd_rn(acc, m.src[i][((size_t)b * m.src_rows[i] + row) * m.channels + c]);
        }
        int orow = m.out_cursor + tt;
        if (orow >= m.out_rows) orow -= m.out_rows;
        m.out[((size_t)b * m.out_rows + orow) * m.channels + c] = __fdiv_rn(acc, (float)m.n);
    }
}

int launch_ring_mean(const RingMeanArgs& m, hipStream_t s) {
    const long long total = (long long)m.batch * m.t * m.channels;
    if (total == 0) return ADK_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ring_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, s, m);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

// ring[b][cursor - h][c] = ring[b][cursor][c] for h = 1..hist  (ReplicationPad1d((hist, 0)), conv_layer.py:190)
__global__ __launch_bounds__(256) void hist_replicate_kernel(float* __restrict__ ring, int rows, int channels, int cursor, int hist, int batch) {
    const long long total = (long long)batch * hist * channels;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int c = (int)(gid % channels);
        const long long bh = gid / channels;
        const int h = (int)(bh % hist) + 1, b = (int)(bh / hist);
        int row = cursor - h;
        if (row < 0) row += rows;
        ring[((size_t)b * rows + row) * channels + c] = ring[((size_t)b * rows + cursor) * channels + c];
    }
}

int launch_hist_replicate(float* ring, int rows, int channels, int cursor, int hist, int batch, hipStream_t s) {
    const long long total = (long long)batch * hist * channels;
    if (total == 0) return ADK_OK;
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(hist_replicate_kernel, dim3((unsigned)blocks), dim3(256), 0, s, ring, rows, channels, cursor, hist, batch);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

}  // namespace adk

using namespace adk;

extern "C" int adk_rvq_encode(const float* z, const float* embed, const float* enorm, int64_t* idx, float* zq,
                              int32_t n_rows, int32_t n_q, int32_t dim, int32_t size, void* stream) {
    if (!z || !embed || !enorm || !idx) return fail(ADK_ERR_ARG, "adk_rvq_encode: null pointer");
    if (n_rows < 0 || n_q <= 0 || dim <= 0 || dim > RVQ_DIM_MAX || size <= 0)
        return fail(ADK_ERR_SHAPE, "adk_rvq_encode: need 0 < dim <= 128, size > 0, n_q > 0");
    if ((reinterpret_cast<uintptr_t>(embed) | reinterpret_cast<uintptr_t>(enorm)) & 15)
        return fail(ADK_ERR_ARG, "adk_rvq_encode: embed/enorm must be 16-byte aligned");
    if (n_rows == 0) return ADK_OK;
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(device_of(z));
    static int variant = -1;                              // ADK_RVQ_V1=1: the first-round kernel, ADK_RVQ_V=2: the round-2 one (A/B and cross-checks)
    static int rb_env = 0;                                // ADK_RVQ_MAXROWS: largest row count the v2 kernel takes (tuning; default 256)
    if (variant < 0) {
        const char* e = getenv("ADK_RVQ_V1"); variant = (e && atoi(e) == 1) ? 1 : 3;
        e = getenv("ADK_RVQ_V"); if (e && variant != 1 && atoi(e) >= 1 && atoi(e) <= 3) variant = atoi(e);
        e = getenv("ADK_RVQ_MAXROWS"); rb_env = e ? atoi(e) : 0;
    }
    rvq_read_env();
    const int rows_opt = g_rvq_rows.load(), v4_min = g_rvq_v4_min.load();
    const int v4_rows = rows_opt == 1 ? (n_rows > 512 ? 4 : 2) : rows_opt;
    if (variant == 3 && v4_rows > 0 && dim == 64 && size == 1024 && n_rows >= v4_min) {
        if (v4_rows == 2)
            hipLaunchKernelGGL(rvq_encode_v4_kernel<2>, dim3((n_rows + 1) / 2), dim3(RVQ_THREADS), 0, s, z, embed, enorm, reinterpret_cast<long long*>(idx), zq, n_rows, n_q);
        else
            hipLaunchKernelGGL(rvq_encode_v4_kernel<4>, dim3((n_rows + 3) / 4), dim3(RVQ_THREADS), 0, s, z, embed, enorm, reinterpret_cast<long long*>(idx), zq, n_rows, n_q);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    }
    if (variant >= 2 && dim == 64 && size == 1024 && n_rows <= (rb_env > 0 ? rb_env : 256)) {
        if (variant == 3) {
            // helper blocks for few rows (see the kernel head; ADK_RVQ_HELPERS per XCD, 0 = none)
            static const int n_help = [] { const char* e = getenv("ADK_RVQ_HELPERS"); const int v = e ? atoi(e) : 2; return v < 0 ? 0 : (v > 8 ? 8 : v); }();
            const unsigned grid3 = (n_help > 0 && n_rows <= 64) ? (unsigned)(((n_rows + 7) & ~7) + 8 * n_help) : (unsigned)n_rows;
            hipLaunchKernelGGL(rvq_encode_v3_kernel, dim3(grid3), dim3(RVQ_THREADS), 0, s, z, embed, enorm,
                               reinterpret_cast<long long*>(idx), zq, n_rows, n_q);
            ADK_HIP_CHECK(hipGetLastError());
            return ADK_OK;
        }
        // the latency kernel: one workgroup per row -- one dispatch round up to 256 rows (measured 37 us for 1..32 rows, 43 us
        // for 256, against 77-78 us of the first-round kernel; at 512 rows = two rounds it only ties, 80 vs 79 us, because each
        // workgroup streams the 2 MB of codes from L2, so the 4-rows-per-workgroup kernel below keeps the large row counts)
        hipLaunchKernelGGL(rvq_encode_v2_kernel<1>, dim3(n_rows), dim3(RVQ_THREADS), 0, s, z, embed, enorm,
                           reinterpret_cast<long long*>(idx), zq, n_rows, n_q);
        ADK_HIP_CHECK(hipGetLastError());
        return ADK_OK;
    }
    constexpr int RB = 4;
    hipLaunchKernelGGL(rvq_encode_kernel<RB>, dim3((n_rows + RB - 1) / RB), dim3(RVQ_THREADS), 0, s, z, embed, enorm,
                       reinterpret_cast<long long*>(idx), zq, n_rows, n_q, dim, size);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

extern "C" int adk_rvq_lookup(const int64_t* idx, const float* codebook, float* zq, int32_t n_rows, int32_t n_q,
                              int32_t dim, int32_t n_codes, void* stream) {
    if (!idx || !codebook || !zq) return fail(ADK_ERR_ARG, "adk_rvq_lookup: null pointer");
    if (n_rows < 0 || n_q <= 0 || dim <= 0 || dim % 4 || n_codes <= 0) return fail(ADK_ERR_SHAPE, "adk_rvq_lookup: bad shape");
    if ((reinterpret_cast<uintptr_t>(codebook) | reinterpret_cast<uintptr_t>(zq)) & 15)
        return fail(ADK_ERR_ARG, "adk_rvq_lookup: codebook/zq must be 16-byte aligned");
    if (n_rows == 0) return ADK_OK;
    DeviceGuard guard(device_of(zq));
    const long long total = (long long)n_rows * (dim / 4);
    long long blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(rvq_lookup_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(idx), codebook, zq, n_rows, n_q, dim, n_codes);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

extern "C" int adk_ring_write(const float* src, adk_ring_view ring, const float* mean, const float* scale,
                              int32_t batch, int32_t t, void* stream) {
    if (!src || !ring.base) return fail(ADK_ERR_ARG, "adk_ring_write: null pointer");
    if ((mean == nullptr) != (scale == nullptr)) return fail(ADK_ERR_ARG, "adk_ring_write: mean and scale go together");
    if (batch < 0 || t < 0 || t > ring.rows || ring.cursor < 0 || ring.cursor >= ring.rows || ring.ch_off != 0)
        return fail(ADK_ERR_SHAPE, "adk_ring_write: bad ring geometry (full rows only: ch_off must be 0)");
    if (batch == 0 || t == 0) return ADK_OK;
    DeviceGuard guard(device_of(ring.base));
    const int src_ch = ring.channels;
    const long long total = (long long)batch * t * src_ch;
    long long blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipLaunchKernelGGL(ring_write_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), src,
                       ring.base, ring.rows, ring.channels, ring.cursor, ring.ch_off, src_ch, mean, scale, batch, t);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

// ---- flag words ----------------------------------------------------------------------------------------------------
namespace adk {
// Round 6: a program's sticky flag word lives in PINNED HOST memory (coherent, mapped into the device); the kernels' rare failure paths
// `atomicOr` into it across PCIe (system memory is uncached on the device side: an atomic op on the bus, which ROCm platforms support), and
// the host reads AND clears it with one atomic exchange of its own -- no kernel, no copy.  Until round 5 the words lived in device memory and
// every read was a 1-thread kernel (exchange -> pinned mirror, ~4 us on the device): one per program and step for the deferred guard, i.e.
// three more launches per pipeline step than the work itself.  An event behind the step now IS the post (adk_program_flags_post).
constexpr int kFlagSlots = 1024;          // programs alive at the same time on one device
struct FlagPool {
    int* host = nullptr;                  // kFlagSlots pinned host words: what the kernels atomicOr into, what the host exchanges
    int* host_dev = nullptr;              // ... their device-side address (what a program's launches are given)
    int* sweep_dev = nullptr;             // one device word + its pinned mirror slot 0: adk_debug_flags' read of the DEVICE-wide word (op-level calls)
    std::vector<int> free_slots;
    int next = 1;
};
static FlagPool g_pool[kMaxDevices];
static std::mutex g_pool_mu;              // slot bookkeeping only: never held across a device synchronisation

static int pool_ready(FlagPool& fp) {     // (g_pool_mu held, the pool's device current)
    if (fp.host) return ADK_OK;
    int* h = nullptr; int* hd = nullptr;
    ADK_HIP_CHECK(hipHostMalloc(reinterpret_cast<void**>(&h), kFlagSlots * sizeof(int), hipHostMallocMapped | hipHostMallocCoherent));
    memset(h, 0, kFlagSlots * sizeof(int));
    ADK_HIP_CHECK(hipHostGetDevicePointer(reinterpret_cast<void**>(&hd), h, 0));
    fp.host = h; fp.host_dev = hd;
    return ADK_OK;
}

int flag_pool_acquire(int device, int** word) {
    std::lock_guard<std::mutex> lk(g_pool_mu);
    FlagPool& fp = g_pool[device];
    const int rc = pool_ready(fp);
    if (rc != ADK_OK) return rc;
    int slot;
    if (!fp.free_slots.empty()) { slot = fp.free_slots.back(); fp.free_slots.pop_back(); }
    else if (fp.next < kFlagSlots) slot = fp.next++;
    else return fail(ADK_ERR_STATE, "more than 1023 live programs on one device");
    __atomic_store_n(fp.host + slot, 0, __ATOMIC_RELEASE);
    *word = fp.host_dev + slot;
    return ADK_OK;
}

static int* pool_host_word(FlagPool& fp, const int* word) {
    if (!fp.host_dev || word <= fp.host_dev || word >= fp.host_dev + kFlagSlots) return nullptr;
    return fp.host + (word - fp.host_dev);
}

void flag_pool_release(int device, int* word) {
    if (!word) return;
    std::lock_guard<std::mutex> lk(g_pool_mu);
    FlagPool& fp = g_pool[device];
    int* h = pool_host_word(fp, word);
    if (!h) return;
    __atomic_store_n(h, 0, __ATOMIC_RELEASE);          // (a released slot reads 0 in the sweep)
    fp.free_slots.push_back((int)(h - fp.host));
}

// read AND clear in one atomic operation: a bit set by a kernel of another HIP stream between a separate read and a separate clear would be lost
int flag_pool_take(int device, int* word, int* v) {
    FlagPool& fp = g_pool[device];
    int* h = pool_host_word(fp, word);
    if (!h) return fail(ADK_ERR_ARG, "flag word outside the device's pool");
    *v = __atomic_exchange_n(h, 0, __ATOMIC_ACQ_REL);
    return ADK_OK;
}

int flag_pool_fetch(int device, int* word, hipStream_t s, int* v) {
    ADK_HIP_CHECK(hipStreamSynchronize(s));            // what was queued on the stream has reported
    return flag_pool_take(device, word, v);
}

__global__ void word_fetch_clear_kernel(int* word, int* out) { *out = atomicExch(word, 0); __threadfence_system(); }

int flag_pool_fetch_all(int device, int* acc) {
    FlagPool& fp = g_pool[device];
    {
        std::lock_guard<std::mutex> lk(g_pool_mu);
        const int rc = pool_ready(fp);
        if (rc != ADK_OK) return rc;
    }
    ADK_HIP_CHECK(hipDeviceSynchronize());             // everything queued on the device has reported
    for (int i = 1; i < kFlagSlots; ++i)
        if (__atomic_load_n(fp.host + i, __ATOMIC_ACQUIRE)) *acc |= __atomic_exchange_n(fp.host + i, 0, __ATOMIC_ACQ_REL);
    // the device-wide word (launches that belong to no program) is device memory: one 1-thread kernel moves it to slot 0 of the pinned words
    hipLaunchKernelGGL(word_fetch_clear_kernel, dim3(1), dim3(1), 0, nullptr, flags_word(), fp.host_dev);
    ADK_HIP_CHECK(hipGetLastError());
    ADK_HIP_CHECK(hipStreamSynchronize(nullptr));
    *acc |= __atomic_exchange_n(fp.host, 0, __ATOMIC_ACQ_REL);
    return ADK_OK;
}

extern "C" int adk_debug_flags(int32_t* out) {
    // OR of the flag words of every device this library has launched on (plus the current one): per device ONE sweep kernel that
    // exchanges every program slot and the device word for 0, after everything queued on that device has finished
    static std::mutex dbg_mu;                          // slot 0 of the host mirror is this function's: one caller at a time
    std::lock_guard<std::mutex> lk(dbg_mu);
    int all = 0;
    const int here = current_device();
    for (int d = 0; d < kMaxDevices; ++d) {
        if (!g_flag_ptr[d] && !g_pool[d].host && d != here) continue;
        DeviceGuard guard(d);
        const int rc = flag_pool_fetch_all(d, &all);
        if (rc != ADK_OK) return rc;
    }
    if (out) *out = all;
    return ADK_OK;
}

}  // namespace adk
