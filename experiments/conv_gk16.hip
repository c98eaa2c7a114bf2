// conv_gk16 -- round 4's DMA-fed 128 x 128-tile split-f16 conv for the wide layers (256 channels per group) whose input ring has a shadow.
// NOT COMPILED, NOT PART OF THE LIBRARY (round 5): it was correct (tests/test_gpu_b256.py of round 4 ran it on every op it supports) and
// 20 % faster than the stream-K launch it replaces when ALONE on the chip (65 -> 29.4 us of span over four measured versions), but a
// workgroup owns its CU (128 KiB of LDS, eight 256-register waves) and inside the three-stream pipeline it cost 2.3 % of the throughput
// (profiles/r4_gk16_timeline.md); it stayed opt-in for a round (ADK_GK16 / adk_set_option("gk16")) and was then moved here with its ABI option
// removed, as round 3 did with its losers.  What it taught went into the product: lane-contiguous stream-K partial tiles and the finished tile
// through LDS (sk_epilogue_lds).  The two fragments below are the kernel (it lived in csrc/conv_mfma.hip behind conv_sk_kernel and uses that
// file's helpers: kSkLoScale, f16x8s, the ConvArgs shadow fields) and its host side (support test, K-part plan, launch).
//
// ================================================================================================
// conv_gk16 -- 128 x 128 tiles, BOTH operands through LDS by LDS-DMA, split-f16 arithmetic (round 4).
//
// The stream-K kernel above gives every wave a 32 x 32 block of its workgroup's 64 x 64 tile: per 64-deep chunk a workgroup requests
// 32 KiB of weight fragments (each wave its own, two waves the same ones) and 16 KiB of activations for 48 MFMAs -- 1 KiB per MFMA.
// For the wide layers (256 channels per group, K up to 2816: the first stage of a v1 vocoder, 5.5 GFLOP per conv at 256 streams)
// that is ~20 TB/s of L2 reads chip-wide for 137 TFLOP/s: the kernel sits on the L2, whatever the loop does (profiles/r2_sk16_analysis.md).
// With a SHADOW ring as input (adk_op_desc.in_shadow: the operand form of act(x) is already in memory) nothing has to pass through
// registers on its way to LDS any more, so the tile can grow to what the accumulators allow:
//   * one workgroup = 4 waves = a 128 x 128 tile of one group, each wave a 64 x 64 quarter (2 x 2 MFMA tiles: 128 accumulator registers
//     for the main and the cross sums); 12 MFMAs per wave and 16-k step on 4 + 4 fragment reads; 64 KiB per chunk for 192 MFMAs --
//     a third of the stream-K kernel's traffic per MFMA;
//   * per 64-deep chunk the workgroup copies 32 KiB of packed weight fragments (lane-linear, as they are) and 32 KiB of shadow rows
//     (128 columns x 256 bytes = the 64 channels of one tap as [8 hi][8 lo] groups) global -> LDS with global_load_lds_dwordx4, 16
//     pieces of 1 KiB per wave, double buffered: the pieces of chunk c + 1 are issued between the MFMA steps of chunk c;
//   * the B image is [column][16 slots of 16 bytes] with slot ^= column & 15 -- the 16 lanes of a ds_read_b128 group sit in 16
//     different columns and would otherwise all hit the same banks; LDS-DMA writes lane-linear, so the swizzle is applied to the
//     per-lane SOURCE address of the copy (cdna_hip_programming.md rule 21);
//   * K is split evenly over `S` workgroups per tile (tiles x S ~ the number of CUs); every part writes its 128 x 128 partial sums
//     write-through to the workspace and bumps the tile's counter; the part that arrives LAST adds all parts in part order -- a fixed
//     order, so the result does not depend on who was last -- runs the epilogue (the stream-K kernel's: bias, residual, output
//     shadow) and puts the counter back to 0.  Nobody ever waits for anybody.
// Per accumulator the order is that of the other split kernels (hi*hi | hi*lo, lo*hi; main + cross / 2048); where K is cut differs
// from the stream-K kernel, so results agree with it to f32 round-off, not bit for bit (as for any two stream-K splits).
struct GkArgs {
    float* ws; unsigned ws_bytes;
    unsigned* counters;   // [tiles]: parts of the tile that have published (0 between launches)
    int S;                // K parts per tile
    int G;                // tiles * S work items
    int m_tiles, n_tiles, nchunks;
    int cpt;              // 64-channel blocks per tap = cin_g / 64
    int kgroups;          // 8-k fragments per 32-row m-tile (K is a multiple of 64 here)
    int mt32_per_g;
    float inv_t_out;
    int* err;
    int dbg;              // ADK_GK16_DBG (tuning; results are garbage): 1 = no B copies after the prologue, 2 = no A copies, 4 = no MFMAs, 8 = no K-part reduction
};

constexpr int GK_BUF = 32 * 1024;                       // 4 * GK_BUF = the dynamic LDS of the kernel: four stage buffers of 16 + 16 KiB; the tail's tile image after the loop
// ADK_GK16_DBG & 16: wall-clock stamps (s_memrealtime, 100 MHz) of wave 0 of every workgroup of the LAST conv_gk16 launch:
// 0 entry, 1 prologue copies issued, 2 first chunk landed (past the first barrier), 3 loop done, 4 slabs published + all parts arrived,
// 5 own slab reduced + finished (epilogue stores issued), 6 exit
__device__ unsigned long long g_gk_trace[512 * 8];
extern "C" int adk_debug_gk_trace(unsigned long long* out, int n) {
    if (n > 512 * 8) n = 512 * 8;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(g_gk_trace), (size_t)n * sizeof(unsigned long long)) == hipSuccess ? 0 : -1;
}
#define GK_STAMP(i) do { if ((gk.dbg & 16) && wave8 == 0) { __builtin_amdgcn_sched_barrier(0); const unsigned long long t_ = __builtin_amdgcn_s_memrealtime(); \
                         if (lane == 0 && r < 512) g_gk_trace[r * 8 + (i)] = t_; __builtin_amdgcn_sched_barrier(0); } } while (0)

#define GK_DMA16(gptr, m0val) do { unsigned m0_keep_; asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" \
                                                            : "=&s"(m0_keep_) : "v"(gptr), "s"(m0val) : "memory"); } while (0)

__global__ __launch_bounds__(512, 1) void conv_gk16_kernel(ConvArgs a, GkArgs gk) {
    extern __shared__ __attribute__((aligned(16))) unsigned char gks[];      // [A0 | A1 | B0 | B1], 32 KiB each (no static __shared__ beside it: that would shift its base off 16 bytes)
    // 8 waves: waves 0-3 multiply (64 x 64 each) and run the tail, waves 4-7 only copy (LDS-DMA) -- two waves per SIMD, one of each kind.
    // A wave that did both stalled at the vector-memory issue while its matrix core idled: the copies alone take 10.7 us of the
    // 128-stream stage-0 conv's loop, the MFMAs + fragment reads alone 12.4 us, one wave doing both 18.2 us (profiles/r4_gk16_timeline.md).
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave8 = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool loader = wave8 >= 4;
    const int wave = wave8 & 3;                             // multiplier: its quarter of the tile; loader: the quarter of the copies it issues
    const int wm = wave >> 1, wn = wave & 1;
    const int l31 = lane & 31, lh = lane >> 5;

    // XCD-contiguous work items: the parts of a tile, and the tiles of a group, share an L2
    const int per_xcd = (gk.G + 7) >> 3;
    const int r = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (r >= gk.G) return;
    GK_STAMP(0);
    const int tile = r / gk.S, part = r - tile * gk.S;
    const int mt = tile % gk.m_tiles;
    const int rest = tile / gk.m_tiles;
    const int nt = rest % gk.n_tiles;
    const int g = rest / gk.n_tiles;
    const int c0 = (part * gk.nchunks) / gk.S, c1 = ((part + 1) * gk.nchunks) / gk.S;

    typedef unsigned char __attribute__((address_space(3)))* lds_u8_t;
    const unsigned lds0 = (unsigned)(size_t)(lds_u8_t)gks;
    const unsigned lane16 = (unsigned)lane * 16u;

    // ---- The K range is walked in STAGES of 32 k (two MFMA k-steps): 16 KiB of weight fragments + 16 KiB of shadow rows per stage,
    // four stage buffers, three stages in flight: a piece has ~2300 MFMA cycles (1.1 us) to land -- what is not in the L2 comes from
    // the Infinity Cache with a first-touch latency of 1-2 us (profiles/r3_load_rate.md); with 64-k chunks and two buffers (one chunk
    // ahead) the loop ran at 1.6 us per chunk against 1.0 us without any copies (profiles/r4_gk16_timeline.md).
    // DMA sources.  A: this wave copies m-tile32 `wave` of the tile: 4 KiB per stage, contiguous in the packed weights ----
    const unsigned char* wsrc = reinterpret_cast<const unsigned char*>(a.wfrag) +
                                ((size_t)(g * gk.mt32_per_g + mt * 4 + wave) * gk.kgroups) * 1024u + lane16;
    // B: piece p of this wave = columns 8 * (4 * wave + p) .. + 7 of the tile, 128 bytes (32 channels: four [8 hi][8 lo] groups) each;
    // lane -> (column, 16-byte slot), slot ^= (column >> 1) & 7 (two columns share a 256-byte bank row: see the reads below)
    const unsigned row_bytes = (unsigned)a.in_ch * 4u;
    const unsigned ring_bytes = (unsigned)a.in_rows * row_bytes;
    const unsigned dil_bytes = (unsigned)a.dilation * row_bytes;
    const unsigned char* bsrc[4];          // column base (stream, group, channel offset) + slot of this lane
    unsigned rowb[4];                      // ring row of tap 0 (bytes)
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const int col = 8 * (4 * wave + p) + (lane >> 3);
        int n = nt * 128 + col;
        if (n >= a.n_total) n = a.n_total - 1;              // (columns past the end: computed on a valid column, never stored)
        const int b = fast_div(n, a.t_out, gk.inv_t_out), t = n - b * a.t_out;
        int row = a.in_row0 + t * a.stride;
        if (row >= a.in_rows) row -= a.in_rows;
        rowb[p] = (unsigned)row * row_bytes;
        const unsigned slot = (unsigned)((lane & 7) ^ ((col >> 1) & 7));
        bsrc[p] = reinterpret_cast<const unsigned char*>(a.in) + (size_t)b * ring_bytes + (size_t)(a.in_choff + g * a.in_gstride) * 4u + slot * 16u;
    }
    constexpr int GK_ST = 16 * 1024;                        // one operand, one stage; buffer q: A at q * 32 KiB, B at q * 32 KiB + 16 KiB
    const int ns = 2 * (c1 - c0);                           // stages of this part
    // (tap, 32-channel block) of the stage whose pieces are issued next: wave-uniform
    int is_ = 0;                                            // ... its index
    int tap_i = c0 / gk.cpt, hblk_i = 2 * (c0 - tap_i * gk.cpt);
    const int hpt = 2 * gk.cpt;                             // 32-channel blocks per tap
    auto issue_stage = [&]() __attribute__((always_inline)) {
        const unsigned buf = (unsigned)(is_ & 3) * 2u * GK_ST;
        if (!(gk.dbg & 2)) {
#pragma unroll
            for (int p = 0; p < 4; ++p)
                GK_DMA16(wsrc + (size_t)(2 * c0 + is_) * 4096u + (size_t)p * 1024u, lds0 + buf + (unsigned)wave * 4096u + (unsigned)p * 1024u);
        }
        if (!(gk.dbg & 1) || is_ < 3) {
#pragma unroll
            for (int p = 0; p < 4; ++p) {
                unsigned rb = rowb[p] + (unsigned)tap_i * dil_bytes;
                if (rb >= ring_bytes) rb -= ring_bytes;
                GK_DMA16(bsrc[p] + rb + (unsigned)hblk_i * 128u, lds0 + buf + GK_ST + (unsigned)(4 * wave + p) * 1024u);
            }
        }
        ++is_;
        if (++hblk_i == hpt) { hblk_i = 0; ++tap_i; }
    };

    if (loader) {
        // ---- the copying waves: stages 0 .. 2 up front (a part has >= 4 stages), then one stage per barrier, three ahead ----
        issue_stage(); issue_stage(); issue_stage();
        for (int sg = 0; sg < ns; ++sg) {
            // my pieces of stage sg have landed (8 per stage; those of the <= 2 stages behind it may stay in flight); past the barrier
            // the multipliers are done reading stage sg - 1, whose buffer the pieces of stage sg + 3 go to
            const int later = ns - 1 - sg;
            if (later >= 2) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
            else if (later == 1) asm volatile("s_waitcnt vmcnt(8)\n\ts_barrier" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)\n\ts_barrier" ::: "memory");
            if (sg + 3 < ns) issue_stage();
        }
        return;                                             // (a wave that has ended is not waited for by the barriers of the tail)
    }

    f32x16 acc[2][2], accx[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) { acc[i][j][e] = 0.f; accx[i][j][e] = 0.f; }
    GK_STAMP(1);

    // this lane's fragment addresses (bytes from the start of a stage buffer)
    const unsigned a_off = (unsigned)(wm * 2) * 4096u + lane16;                           // + i * 4096 + (2 * st + half) * 1024
    const unsigned x8 = (unsigned)((l31 >> 1) & 7);
    const unsigned b_off = GK_ST + (unsigned)(wn * 64 + l31) * 128u;                      // + jn * 32 * 128 + slot * 16

    for (int sg = 0; sg < ns; ++sg) {
        // stage sg has landed (the copying waves waited for their pieces before they arrived here); my fragment reads of stage sg - 1
        // have returned, so its buffer may be overwritten
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
        if (sg == 0) GK_STAMP(2);
        const unsigned char* Sb = gks + (size_t)(sg & 3) * 2 * GK_ST;
        if (gk.dbg & 4) continue;
#pragma unroll
        for (int st = 0; st < 2; ++st) {
            f16x8s ah[2], al[2], bh[2], bl[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                ah[i] = *reinterpret_cast<const f16x8s*>(Sb + a_off + i * 4096 + (2 * st) * 1024);
                al[i] = *reinterpret_cast<const f16x8s*>(Sb + a_off + i * 4096 + (2 * st + 1) * 1024);
            }
            const unsigned hs = ((unsigned)(4 * st + 2 * lh) ^ x8) * 16u, ls = ((unsigned)(4 * st + 2 * lh + 1) ^ x8) * 16u;
#pragma unroll
            for (int j = 0; j < 2; ++j) {
                bh[j] = *reinterpret_cast<const f16x8s*>(Sb + b_off + j * 32 * 128 + hs);
                bl[j] = *reinterpret_cast<const f16x8s*>(Sb + b_off + j * 32 * 128 + ls);
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
    GK_STAMP(3);
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = fmaf(accx[i][j][e], kSkLoInv, acc[i][j][e]);

    // ---- tail.  The tile goes through LDS (the operand buffers are free now) as T[column][128 channels], 528-byte rows: from there
    // every global access of the tail is a full, coalesced 16 bytes per lane along the channel axis (the accumulator layout has a lane's
    // neighbours 32 columns = 32 ring rows apart).  With S > 1 the parts of a tile exchange through the workspace, reduce-scatter:
    // part q finishes the columns [q * 128 / S, (q + 1) * 128 / S) -- it publishes the other parts' column slabs (write-through), counts
    // itself in, waits until all S parts have (they were dispatched back to back and run in step: they arrive within ~1 us of each
    // other; the wait is bounded, device flag bit 1), adds the S contributions to ITS slab in part order -- a fixed order, whoever
    // arrives when -- and runs the epilogue on it.  The last part to have read puts the tile's two counters back to 0. ----
    constexpr int TS = 528;
    __syncthreads();                                        // every wave is done with the operand buffers
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            unsigned char* trow = gks + (size_t)(wn * 64 + j * 32 + l31) * TS + (size_t)(wm * 64 + i * 32 + 4 * lh) * 4;
#pragma unroll
            for (int qd = 0; qd < 4; ++qd)
                *reinterpret_cast<float4*>(trow + qd * 32) = make_float4(acc[i][j][4 * qd], acc[i][j][4 * qd + 1], acc[i][j][4 * qd + 2], acc[i][j][4 * qd + 3]);
        }
    __syncthreads();
    const int S = (gk.dbg & 8) ? 1 : gk.S;
    const int W = 128 / S;                                  // columns of this part's slab (S is a power of two <= 8)
    const int my0 = (gk.dbg & 8) ? 0 : part * W;
    const __amdgpu_buffer_rsrc_t rsrc_ws = __builtin_amdgcn_make_buffer_rsrc(gk.ws, 0, gk.ws_bytes, 0x00020000);
    const unsigned slab_bytes = (unsigned)W * 512u;
    if (S > 1) {
        // my contribution to the OTHER parts' slabs: 8 columns (4 KiB) per pass, a wave = two whole 512-byte columns per store
        for (int q = 0; q < S; ++q) {
            if (q == part) continue;
            const unsigned dst = ((unsigned)((tile * S + q) * S + part)) * slab_bytes;
            for (int c = tid >> 5; c < W; c += 8) {
                const u32x4 v = *reinterpret_cast<const u32x4*>(gks + (size_t)(q * W + c) * TS + (size_t)(tid & 31) * 16);
                __builtin_amdgcn_raw_buffer_store_b128(v, rsrc_ws, dst + (unsigned)c * 512u + (unsigned)(tid & 31) * 16u, 0, 16 /* sc1 */);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid == 0) {
            __hip_atomic_fetch_add(gk.counters + 2 * tile, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned spins = 0;
            while (__hip_atomic_load(gk.counters + 2 * tile, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned)S) {
                __builtin_amdgcn_s_sleep(2);
                if (++spins > (1u << 20)) { atomicOr(gk.err, 2); break; }              // never hang the device
            }
        }
        __syncthreads();
        GK_STAMP(4);
    }
    // ---- my slab: 16 columns per pass, a thread = 8 channels (one shadow group) of one column ----
    const int cg = tid & 15;
    const int ml = mt * 128 + 8 * cg;                       // channel within the group
    const int mg = g * a.cout_g + ml;
    float4 bias0 = make_float4(0.f, 0.f, 0.f, 0.f), bias1 = bias0;
    if (a.bias) { bias0 = *reinterpret_cast<const float4*>(a.bias + mg); bias1 = *reinterpret_cast<const float4*>(a.bias + mg + 4); }
    int ph = 0, ocol = mg;
    if (a.up > 1) { ph = mg / a.cout_real; ocol = mg - ph * a.cout_real; }
    bool bad = false;
    for (int c = tid >> 4; c < W; c += 16) {
        const int n = nt * 128 + my0 + c;
        float4 t0 = make_float4(0.f, 0.f, 0.f, 0.f), t1 = t0;
        u32x4 pv[2 * 8];
        if (S > 1) {
#pragma unroll
            for (int sp = 0; sp < 8; ++sp)
                if (sp < S && sp != part) {
                    const unsigned src = ((unsigned)((tile * S + part) * S + sp)) * slab_bytes + (unsigned)c * 512u + (unsigned)cg * 32u;
                    pv[2 * sp] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, src, 0, 16 /* sc1 */);
                    pv[2 * sp + 1] = __builtin_amdgcn_raw_buffer_load_b128(rsrc_ws, src + 16u, 0, 16 /* sc1 */);
                }
        }
        const float4 own0 = *reinterpret_cast<const float4*>(gks + (size_t)(my0 + c) * TS + (size_t)cg * 32);
        const float4 own1 = *reinterpret_cast<const float4*>(gks + (size_t)(my0 + c) * TS + (size_t)cg * 32 + 16);
#pragma unroll
        for (int sp = 0; sp < 8; ++sp)
            if (sp < S) {
                if (sp == part || S == 1) {
                    t0.x += own0.x; t0.y += own0.y; t0.z += own0.z; t0.w += own0.w;
                    t1.x += own1.x; t1.y += own1.y; t1.z += own1.z; t1.w += own1.w;
                } else {
                    const u32x4 u = pv[2 * sp], v = pv[2 * sp + 1];
                    t0.x += __uint_as_float(u.x); t0.y += __uint_as_float(u.y); t0.z += __uint_as_float(u.z); t0.w += __uint_as_float(u.w);
                    t1.x += __uint_as_float(v.x); t1.y += __uint_as_float(v.y); t1.z += __uint_as_float(v.z); t1.w += __uint_as_float(v.w);
                }
            }
        if (n >= a.n_total) continue;
        // the stream-K kernel's epilogue (sk_epilogue), 8 channels of one column at a time: bias, residual, output activation, store, shadow
        bad |= !(fabsf(t0.x) <= 3.0e38f) | !(fabsf(t0.y) <= 3.0e38f) | !(fabsf(t0.z) <= 3.0e38f) | !(fabsf(t0.w) <= 3.0e38f) |
               !(fabsf(t1.x) <= 3.0e38f) | !(fabsf(t1.y) <= 3.0e38f) | !(fabsf(t1.z) <= 3.0e38f) | !(fabsf(t1.w) <= 3.0e38f);
        const int bb = fast_div(n, a.t_out, gk.inv_t_out), t = n - bb * a.t_out;
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
            f16x8s hi, lo;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float y = act_apply(x[e], a.sh_act, a.sh_slope);
                const _Float16 h = (_Float16)y;
                hi[e] = h;
                lo[e] = (_Float16)((y - (float)h) * kSkLoScale);
            }
            unsigned char* sp_ = reinterpret_cast<unsigned char*>(a.out_sh + ((size_t)bb * a.out_rows + orow) * a.out_ch + a.out_choff) + (size_t)(ocol >> 3) * 32;
            *reinterpret_cast<f16x8s*>(sp_) = hi;
            *reinterpret_cast<f16x8s*>(sp_ + 16) = lo;
        }
    }
    if (bad) atomicOr(gk.err, 8);
    if (S > 1) {
        GK_STAMP(5);
        __syncthreads();                                    // every thread of this part has read the other parts' slabs
        if (tid == 0) {
            const unsigned gone = __hip_atomic_fetch_add(gk.counters + 2 * tile + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if (gone == (unsigned)(S - 1)) {                // the last part to leave: all S are past their waits and their reads
                __hip_atomic_store(gk.counters + 2 * tile, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                __hip_atomic_store(gk.counters + 2 * tile + 1, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
    }
    GK_STAMP(6);
}


// ---- conv_gk16 host side ----
constexpr int kGkCounters = 1024;
static int g_gk = -1;          // ADK_GK16 / adk_set_option("gk16"): 0 = never (default), 1 = where it is preferred, 2 = wherever it is supported (tests).
                               // Default off: alone on the chip the kernel takes the wide stage-0 convs from 37.5 to 29.5 us (rocprof: 34.5 vs 41 by events), but a
                               // workgroup owns its CU (128 KiB of LDS, 8 waves of 256 registers): in the three-stream pipeline, where the other programs'
                               // workgroups fill the gaps of the stream-K launches, it costs 2.3 % of the throughput (profiles/r4_gk16_timeline.md)
static int g_gk_min_work = 0;  // ADK_GK16_MIN_WORK: tiles * chunks from which the kernel is preferred

bool conv_gk16_supported(const ConvArgs& a) {
    auto al16 = [](const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; };
    if (!a.in_sh || !a.wfrag || !al16(a.in_sh) || !al16(a.wfrag) || !al16(a.out)) return false;
    if (a.cin_g % 64 || a.cout_g % 128 || a.cout_real % 4 || a.n_total < 1 || a.n_total >= (1 << 24)) return false;
    if (a.in_ch % 8 || a.in_choff % 8 || a.in_gstride % 8 || a.out_ch % 4 || a.out_choff % 4) return false;       // whole 8-channel shadow groups
    if ((unsigned long long)a.batch * a.in_rows * a.in_ch * 4ull >= 0x80000000ull) return false;
    if (a.bias && !al16(a.bias)) return false;
    if (a.res && (a.res_ch % 4 || a.res_choff % 4 || a.res_gstride % 4 || !al16(a.res))) return false;
    const long long tiles = (long long)(a.cout_g / 128) * ((a.n_total + 127) / 128) * a.groups;
    return 2 * tiles <= kGkCounters;          // two counters per tile (arrived / left)
}

static void gk_read_env() {
    if (g_gk < 0) {
        const char* e = getenv("ADK_GK16"); g_gk = e ? atoi(e) : 0;
        e = getenv("ADK_GK16_MIN_WORK"); g_gk_min_work = e ? atoi(e) : 1536;
    }
}
void conv_gk16_mode(int mode) { gk_read_env(); g_gk = mode < 0 ? 0 : mode; }      // adk_set_option("gk16", ...)

bool conv_gk16_preferred(const ConvArgs& a) {
    gk_read_env();
    if (!g_gk || !conv_gk16_supported(a)) return false;
    if (g_gk >= 2) return true;
    // enough work for ~240 workgroups of >= 6 chunks: the wide grouped convs of a v1 vocoder's first stage at >= 128 streams
    // (60 tiles x 44 chunks at 256); the smaller stream-K launches (a few tiles, K <= 1792) stay where they are
    const long long tiles = (long long)(a.cout_g / 128) * ((a.n_total + 127) / 128) * a.groups;
    return tiles * (a.ktot / 64) >= g_gk_min_work;
}

int launch_conv_gk16(const ConvArgs& a, hipStream_t s, Workspace& ws) {
    if (a.n_total == 0) return ADK_OK;
    if (!conv_gk16_supported(a)) return fail(ADK_ERR_STATE, "conv_gk16: unsupported arguments");
    GkArgs gk;
    gk.m_tiles = a.cout_g / 128;
    gk.n_tiles = (a.n_total + 127) / 128;
    gk.nchunks = a.ktot / 64;
    gk.cpt = a.cin_g / 64;
    gk.kgroups = gk.nchunks * 8;
    gk.mt32_per_g = a.cout_g / 32;
    gk.inv_t_out = 1.0f / (float)a.t_out;
    const int tiles = gk.m_tiles * gk.n_tiles * a.groups;
    int S = 1;                                            // K parts per tile: a power of two <= 8, one workgroup per CU, >= 2 chunks per part
    while (S < 8 && tiles * S * 2 <= 256 && gk.nchunks / (S * 2) >= 2) S *= 2;
    gk.S = S; gk.G = tiles * S;
    size_t flags_offset = 0;
    const size_t need = conv_mfma_workspace_bytes(&flags_offset);
    if (!ws.ptr || ws.bytes < need || (size_t)gk.G * 65536 > flags_offset) return fail(ADK_ERR_STATE, "conv_gk16: workspace missing or too small");
    gk.ws = ws.ptr; gk.ws_bytes = (unsigned)std::min<size_t>(flags_offset, 0x7fffffffu);
    gk.counters = reinterpret_cast<unsigned*>(reinterpret_cast<char*>(ws.ptr) + need - kGkCounters * sizeof(unsigned));
    gk.err = conv_err_word(a);
    static int dbg = -1;
    if (dbg < 0) { const char* e = getenv("ADK_GK16_DBG"); dbg = e ? atoi(e) : 0; }
    gk.dbg = dbg;
    constexpr size_t lds = 4 * GK_BUF;
    static bool attr_dev[kMaxDevices] = {};
    bool& attr = attr_dev[current_device()];
    if (!attr) {
        ADK_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_gk16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = true;
    }
    ConvArgs b = a;
    b.in = a.in_sh;
    const unsigned grid = (unsigned)((gk.G + 7) / 8 * 8);
    hipLaunchKernelGGL(conv_gk16_kernel, dim3(grid), dim3(512), lds, s, b, gk);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

