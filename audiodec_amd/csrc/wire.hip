// Bit-packed code wire format (the step between transmitter and receiver that the reference leaves
// as an int64 tensor in a queue.Queue, bin/stream.py:224,230).
//
// One frame of one stream = n_q codes of `bits` bits each (10 for 1024-entry codebooks), packed
// LSB-first into ceil(n_q*bits/8) bytes: code q occupies bits [q*bits, (q+1)*bits).  48 kHz hop 300
// with 8 codebooks -> 80 bit = 10 bytes per frame = 12.8 kbps, the bitrate the reference quotes
// (README.md:6).  The stage offset `size*q` of the emitted indices (vq_module.py:145-146) is implicit.
#include "adk_common.h"

namespace adk {

__device__ __forceinline__ unsigned long long code_of(const long long* idx, int q, int row, int n_rows, int size) {
    return (unsigned long long)(idx[(size_t)q * n_rows + row] - (long long)size * q);
}

// one thread per output byte
__global__ __launch_bounds__(256) void codes_pack_kernel(const long long* __restrict__ idx, unsigned char* __restrict__ out,
                                                         int n_rows, int n_q, int bits, int size, int frame_bytes, int* flags) {
    const long long total = (long long)n_rows * frame_bytes;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(gid / frame_bytes), byte = (int)(gid - (long long)row * frame_bytes);
        const int b0 = byte * 8;                       // first bit of this byte
        unsigned v = 0;
        int q = b0 / bits;
        while (q < n_q && q * bits < b0 + 8) {
            unsigned long long c = code_of(idx, q, row, n_rows, size);
            if (c >= (unsigned long long)size) { atomicOr(flags, 4); c = 0; }   // not a code of stage q: flag it and keep it out of
                                                                               // the neighbouring codes' bits (packs as code 0)
            const int shift = q * bits - b0;           // position of the code's bit 0 relative to this byte
            v |= shift >= 0 ? (unsigned)((c << shift) & 0xffull) : (unsigned)((c >> (-shift)) & 0xffull);
            ++q;
        }
        out[gid] = (unsigned char)v;
    }
}

__device__ __forceinline__ unsigned unpack_code(const unsigned char* frame, int q, int bits, int frame_bytes) {
    const int b0 = q * bits;
    unsigned long long w = 0;
    const int first = b0 >> 3;
#pragma unroll
    for (int k = 0; k < 4; ++k)
        if (first + k < frame_bytes) w |= (unsigned long long)frame[first + k] << (8 * k);
    return (unsigned)((w >> (b0 & 7)) & ((1ull << bits) - 1ull));
}

__global__ __launch_bounds__(256) void codes_unpack_kernel(const unsigned char* __restrict__ in, long long* __restrict__ idx,
                                                           int n_rows, int n_q, int bits, int size, int frame_bytes) {
    const long long total = (long long)n_rows * n_q;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int q = (int)(gid / n_rows), row = (int)(gid - (long long)q * n_rows);
        idx[gid] = (long long)unpack_code(in + (size_t)row * frame_bytes, q, bits, frame_bytes) + (long long)size * q;
    }
}

// unpack fused into ResidualVQ.lookup (layers/vq_module.py:159-161): zq[row] = sum_q codebook[size*q + code_q]
__global__ __launch_bounds__(256) void codes_lookup_kernel(const unsigned char* __restrict__ in, const float* __restrict__ codebook,
                                                           float* __restrict__ zq, int n_rows, int n_q, int bits, int size,
                                                           int frame_bytes, int dim, int* flags) {
    const int d4 = dim / 4;
    const long long total = (long long)n_rows * d4;
    for (long long gid = (long long)blockIdx.x * blockDim.x + threadIdx.x; gid < total; gid += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(gid / d4), c = (int)(gid - (long long)row * d4);
        const unsigned char* frame = in + (size_t)row * frame_bytes;
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int q = 0; q < n_q; ++q) {
            unsigned code = unpack_code(frame, q, bits, frame_bytes);
            if (code >= (unsigned)size) { atomicOr(flags, 1); code = 0; }
            const float4 e = *reinterpret_cast<const float4*>(codebook + ((size_t)q * size + code) * dim + 4 * c);
            s.x = __fadd_rn(s.x, e.x); s.y = __fadd_rn(s.y, e.y); s.z = __fadd_rn(s.z, e.z); s.w = __fadd_rn(s.w, e.w);
        }
        *reinterpret_cast<float4*>(zq + (size_t)row * dim + 4 * c) = s;
    }
}

static unsigned grid_for(long long total) {
    long long blocks = (total + 255) / 256;
    return (unsigned)(blocks > 4096 ? 4096 : (blocks < 1 ? 1 : blocks));
}

}  // namespace adk

using namespace adk;

extern "C" int32_t adk_codes_frame_bytes(int32_t n_q, int32_t bits) {
    if (n_q <= 0 || bits <= 0 || bits > 24) return -1;
    return (n_q * bits + 7) / 8;
}

static int check_wire(const char* who, int n_rows, int n_q, int bits, int size) {
    if (n_rows < 0 || n_q <= 0 || bits <= 0 || bits > 24 || size <= 0 || size > (1 << bits))
        return fail(ADK_ERR_SHAPE, std::string(who) + ": need 0 < bits <= 24, 0 < size <= 2^bits, n_q > 0");
    return ADK_OK;
}

extern "C" int adk_codes_pack(const int64_t* idx, uint8_t* out, int32_t n_rows, int32_t n_q, int32_t bits, int32_t size, void* stream) {
    int rc = check_wire("adk_codes_pack", n_rows, n_q, bits, size);
    if (rc != ADK_OK || n_rows == 0) return rc;              // an empty batch of frames is fine (and has no pointers)
    if (!idx || !out) return fail(ADK_ERR_ARG, "adk_codes_pack: null pointer");
    const int fb = (n_q * bits + 7) / 8;
    DeviceGuard guard(device_of(out));
    hipLaunchKernelGGL(codes_pack_kernel, dim3(grid_for((long long)n_rows * fb)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       reinterpret_cast<const long long*>(idx), out, n_rows, n_q, bits, size, fb, flags_word());
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

extern "C" int adk_codes_unpack(const uint8_t* in, int64_t* idx, int32_t n_rows, int32_t n_q, int32_t bits, int32_t size, void* stream) {
    int rc = check_wire("adk_codes_unpack", n_rows, n_q, bits, size);
    if (rc != ADK_OK || n_rows == 0) return rc;
    if (!idx || !in) return fail(ADK_ERR_ARG, "adk_codes_unpack: null pointer");
    const int fb = (n_q * bits + 7) / 8;
    DeviceGuard guard(device_of(idx));
    hipLaunchKernelGGL(codes_unpack_kernel, dim3(grid_for((long long)n_rows * n_q)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       in, reinterpret_cast<long long*>(idx), n_rows, n_q, bits, size, fb);
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}

extern "C" int adk_codes_lookup(const uint8_t* in, const float* codebook, float* zq, int32_t n_rows, int32_t n_q, int32_t bits,
                                int32_t size, int32_t dim, void* stream) {
    int rc = check_wire("adk_codes_lookup", n_rows, n_q, bits, size);
    if (rc != ADK_OK) return rc;
    if (dim <= 0 || dim % 4) return fail(ADK_ERR_SHAPE, "adk_codes_lookup: dim % 4 != 0");
    if (n_rows == 0) return ADK_OK;
    if (!in || !codebook || !zq) return fail(ADK_ERR_ARG, "adk_codes_lookup: null pointer");
    if ((reinterpret_cast<uintptr_t>(codebook) | reinterpret_cast<uintptr_t>(zq)) & 15)
        return fail(ADK_ERR_ARG, "adk_codes_lookup: codebook/zq must be 16-byte aligned");
    if (n_rows == 0) return ADK_OK;
    const int fb = (n_q * bits + 7) / 8;
    DeviceGuard guard(device_of(zq));
    hipLaunchKernelGGL(codes_lookup_kernel, dim3(grid_for((long long)n_rows * (dim / 4))), dim3(256), 0, static_cast<hipStream_t>(stream),
                       in, codebook, zq, n_rows, n_q, bits, size, fb, dim, flags_word());
    ADK_HIP_CHECK(hipGetLastError());
    return ADK_OK;
}
