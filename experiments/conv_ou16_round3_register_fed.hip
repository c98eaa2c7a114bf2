// conv_ou16_kernel -- the register-fed form of round 3 (every lane loads the B fragments of ITS time step straight from the ring, 24 x 16 B
// in flight per lane; both weight sets by LDS-DMA).  NOT COMPILED, NOT PART OF THE LIBRARY since round 6: the LDS-DMA-fed kernel in
// audiodec_amd/csrc/conv_ou16.hip replaced it -- 12.7 -> 10.0 us per 256-stream launch by the in-kernel clock, 10.9 -> 8.0 us for one stream,
// bit-identical outputs (profiles/r6_ou16_timeline.md).  It lived in that file in front of the round-6 kernel and uses its helpers
// (OuArgs, ou_act, OU_STAMP, the OU_* constants); `ADK_OU16_V=1` selected it for the A/B sessions experiments/sessions/r6_s1.sh ... r6_s3.sh.
// KS1 = 16-channel chunks of the 1x1 conv's input (192 channels: 12); MT2 = 32-row tiles of the transposed conv's GEMM (s * Cout / 32)
template <int ACT, int KS1, int MT2>
__global__ __launch_bounds__(256, 1) void conv_ou16_kernel(ConvArgs a1, ConvArgs a2, OuArgs u) {
    extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int l31 = lane & 31, lh = lane >> 5;
    const int b = blockIdx.x;
    const int T = a1.t_out;
    const int t = wave * 32 + l31;
    const bool valid = t < T;
    const int tt = valid ? t : T - 1;                     // padded columns work on a copy of the last one; nothing of theirs is stored
    OU_STAMP(0);

    const int w1_bytes = 2 * u.ks1p * 2048, w2_bytes = MT2 * OU_KS2 * 2048;
    unsigned char* w1l = lds;                              // [2 m-tiles][ks1p chunks][hi | lo][64 lanes][16 B]
    unsigned char* w2l = lds + w1_bytes;                   // [MT2][8 chunks][hi | lo][64 lanes][16 B]
    unsigned char* cbuf = w2l + w2_bytes;                  // [1 + OU_TMAX rows][OU_RSC]: row 0 = c[-1], row 1 + t = c[t]
    float* b2l = reinterpret_cast<float*>(cbuf + (1 + OU_TMAX) * OU_RSC);   // [32 * MT2]
    float* b1l = b2l + 32 * MT2;                           // [64]

    // ---- loads of this lane, oldest first: biases, the history row of c, the B fragments of the 1x1 conv ----
    float bias2_v = 0.f, bias1_v = 0.f;
    if (a2.bias && tid < 32 * MT2) bias2_v = a2.bias[tid];
    if (a1.bias && tid < OU_CM) bias1_v = a1.bias[tid];
    float4 hrow = make_float4(0.f, 0.f, 0.f, 0.f);         // c[-1]: what the previous call left in front of the cursor of the 64-channel ring
    if (tid < OU_CM / 4) hrow = *reinterpret_cast<const float4*>(a2.in + ((size_t)b * a2.in_rows + a2.in_row0) * a2.in_ch + a2.in_choff + 4 * tid);
    float4 xr[KS1][2];
    {
        int row = a1.in_row0 + tt;
        if (row >= a1.in_rows) row -= a1.in_rows;
        const float4* p = reinterpret_cast<const float4*>(a1.in + ((size_t)b * a1.in_rows + row) * a1.in_ch + a1.in_choff + 8 * lh);
#pragma unroll
        for (int s = 0; s < KS1; ++s) { xr[s][0] = p[4 * s]; xr[s][1] = p[4 * s + 1]; }
    }
    // ---- both weight sets: lane-linear copies global -> LDS (LDS-DMA, no registers), in the same round trip ----
    {
        const unsigned char* g1 = reinterpret_cast<const unsigned char*>(a1.wfrag) + (size_t)tid * 16;
        const unsigned char* g2 = reinterpret_cast<const unsigned char*>(a2.wfrag) + (size_t)tid * 16;
        unsigned char* l1 = w1l + wave * 1024;             // wave-uniform base; the lane offset is implicit
        unsigned char* l2 = w2l + wave * 1024;
        for (int i = 0; i < w1_bytes / 4096; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(g1 + 4096 * i), (lptr_t)(l1 + 4096 * i), 16, 0, 0);
#pragma unroll
        for (int i = 0; i < MT2 * OU_KS2 * 2048 / 4096; ++i)
            __builtin_amdgcn_global_load_lds((gptr_t)(g2 + 4096 * i), (lptr_t)(l2 + 4096 * i), 16, 0, 0);
    }
    OU_STAMP(1);
    __builtin_amdgcn_sched_barrier(0);
    if (tid < 32 * MT2) b2l[tid] = bias2_v;
    if (tid < OU_CM) b1l[tid] = bias1_v;
    if (tid < OU_CM / 4) {                                 // history row: activation, split, into row 0 of the c buffer
        const float x[4] = {hrow.x, hrow.y, hrow.z, hrow.w};
        f16x4u hi, lo;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float v = ou_act<ACT>(x[e], a2.slope);
            const _Float16 h = (_Float16)v;
            hi[e] = h; lo[e] = (_Float16)((v - (float)h) * kOuLoScale);
        }
        *reinterpret_cast<f16x4u*>(cbuf + 8 * tid) = hi;
        *reinterpret_cast<f16x4u*>(cbuf + 2 * OU_CM + 8 * tid) = lo;
    }
    // this wave's activation loads and its slices of W1 have landed (the other waves read them); the MT2 * 4 LDS-DMA pieces of W2,
    // issued last, may still be in flight: they are only needed after GEMM 1 (timeline: profiles/r3_ou16_timeline.md)
    if constexpr (MT2 == 3) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
    else if constexpr (MT2 == 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    __syncthreads();
    OU_STAMP(2);

    bool bad = false;
    // ---- GEMM 1: c[m][t] = sum_k W1[m][k] x[k][t], 64 rows (two m-tiles) x this wave's 32 steps ----
    {
        f32x16 am[2], ac[2];
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int e = 0; e < 16; ++e) { am[mt][e] = 0.f; ac[mt][e] = 0.f; }
#pragma unroll
        for (int s = 0; s < KS1; ++s) {
            const float x[8] = {xr[s][0].x, xr[s][0].y, xr[s][0].z, xr[s][0].w, xr[s][1].x, xr[s][1].y, xr[s][1].z, xr[s][1].w};
            f16x8u bh, bl;
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const _Float16 h = (_Float16)x[e];
                bh[e] = h; bl[e] = (_Float16)((x[e] - (float)h) * kOuLoScale);
            }
#pragma unroll
            for (int mt = 0; mt < 2; ++mt) {
                const unsigned char* wp = w1l + (size_t)(mt * u.ks1p + s) * 2048 + lane * 16;
                const f16x8u Ah = *reinterpret_cast<const f16x8u*>(wp);
                const f16x8u Al = *reinterpret_cast<const f16x8u*>(wp + 1024);
                am[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh, am[mt], 0, 0, 0);
                ac[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl, ac[mt], 0, 0, 0);
                ac[mt] = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh, ac[mt], 0, 0, 0);
            }
        }
        OU_STAMP(3);
        // c (+ bias): the last step's row goes to the 64-channel ring (the next call's history); act(c), split, to the LDS buffer
        float* crow = nullptr;
        if (valid && t == T - 1) {
            int row = a1.out_cursor + t;
            if (row >= a1.out_rows) row -= a1.out_rows;
            crow = a1.out + ((size_t)b * a1.out_rows + row) * a1.out_ch + a1.out_choff;
        }
        unsigned char* lrow = cbuf + (1 + t) * OU_RSC;
#pragma unroll
        for (int mt = 0; mt < 2; ++mt)
#pragma unroll
            for (int qd = 0; qd < 4; ++qd) {
                const int ml = mt * 32 + 8 * qd + 4 * lh;
                float v[4] = {fmaf(ac[mt][4 * qd], kOuLoInv, am[mt][4 * qd]), fmaf(ac[mt][4 * qd + 1], kOuLoInv, am[mt][4 * qd + 1]),
                              fmaf(ac[mt][4 * qd + 2], kOuLoInv, am[mt][4 * qd + 2]), fmaf(ac[mt][4 * qd + 3], kOuLoInv, am[mt][4 * qd + 3])};
                if (valid) bad |= !(fabsf(v[0]) <= 3.0e38f) | !(fabsf(v[1]) <= 3.0e38f) | !(fabsf(v[2]) <= 3.0e38f) | !(fabsf(v[3]) <= 3.0e38f);
                if (a1.bias) {
                    const float4 bb = *reinterpret_cast<const float4*>(b1l + ml);
                    v[0] += bb.x; v[1] += bb.y; v[2] += bb.z; v[3] += bb.w;
                }
                if (crow) *reinterpret_cast<float4*>(crow + ml) = make_float4(v[0], v[1], v[2], v[3]);
                if (valid) {
                    f16x4u hi, lo;
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float y = ou_act<ACT>(v[e], a2.slope);
                        const _Float16 h = (_Float16)y;
                        hi[e] = h; lo[e] = (_Float16)((y - (float)h) * kOuLoScale);
                    }
                    *reinterpret_cast<f16x4u*>(lrow + 2 * ml) = hi;
                    *reinterpret_cast<f16x4u*>(lrow + 2 * OU_CM + 2 * ml) = lo;
                }
            }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");       // W2 has landed
    __syncthreads();
    OU_STAMP(4);

    // ---- GEMM 2: the polyphase transposed conv; k = (tap j, channel), tap 0 = the older row c[t-1] = buffer row t, tap 1 = row t + 1 ----
    float* outb = a2.out + (size_t)b * a2.out_rows * a2.out_ch + a2.out_choff;
    int orow0 = a2.out_cursor + t * a2.up;
    orow0 %= a2.out_rows;
    const unsigned char* xb = cbuf + tt * OU_RSC + 16 * lh;
    f16x8u bh[OU_KS2], bl[OU_KS2];
#pragma unroll
    for (int s = 0; s < OU_KS2; ++s) {
        const unsigned char* p = xb + (s / 4) * OU_RSC + 32 * (s % 4);
        bh[s] = *reinterpret_cast<const f16x8u*>(p);
        bl[s] = *reinterpret_cast<const f16x8u*>(p + 2 * OU_CM);
    }
#pragma unroll
    for (int mt = 0; mt < MT2; ++mt) {
        f32x16 am, ac;
#pragma unroll
        for (int e = 0; e < 16; ++e) { am[e] = 0.f; ac[e] = 0.f; }
        const unsigned char* wp = w2l + (size_t)mt * OU_KS2 * 2048 + lane * 16;
#pragma unroll
        for (int s = 0; s < OU_KS2; ++s) {
            const f16x8u Ah = *reinterpret_cast<const f16x8u*>(wp + s * 2048);
            const f16x8u Al = *reinterpret_cast<const f16x8u*>(wp + s * 2048 + 1024);
            am = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bh[s], am, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Ah, bl[s], ac, 0, 0, 0);
            ac = __builtin_amdgcn_mfma_f32_32x32x16_f16(Al, bh[s], ac, 0, 0, 0);
        }
        if (!valid) continue;
#pragma unroll
        for (int qd = 0; qd < 4; ++qd) {
            const int ml = mt * 32 + 8 * qd + 4 * lh;       // GEMM row = phase * cout_real + co
            float4 v = make_float4(fmaf(ac[4 * qd], kOuLoInv, am[4 * qd]), fmaf(ac[4 * qd + 1], kOuLoInv, am[4 * qd + 1]),
                                   fmaf(ac[4 * qd + 2], kOuLoInv, am[4 * qd + 2]), fmaf(ac[4 * qd + 3], kOuLoInv, am[4 * qd + 3]));
            bad |= !(fabsf(v.x) <= 3.0e38f) | !(fabsf(v.y) <= 3.0e38f) | !(fabsf(v.z) <= 3.0e38f) | !(fabsf(v.w) <= 3.0e38f);
            const float4 bb = *reinterpret_cast<const float4*>(b2l + ml);
            v.x += bb.x; v.y += bb.y; v.z += bb.z; v.w += bb.w;
            const int ph = (int)(((float)ml + 0.5f) * u.inv_cout_real);        // ml / cout_real, exact for these sizes
            int r2 = orow0 + ph;
            if (r2 >= a2.out_rows) r2 -= a2.out_rows;
            *reinterpret_cast<float4*>(outb + (size_t)r2 * a2.out_ch + (ml - ph * a2.cout_real)) = v;
        }
    }
    OU_STAMP(5);
    if (bad) atomicOr(u.err, 8);
}


