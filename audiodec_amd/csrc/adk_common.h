// Internal helpers shared by the HIP translation units of libaudiodec_hip.so (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <string>
#include <algorithm>
#include <cstring>
#include "../../include/audiodec_hip.h"

namespace adk {

void set_error(const std::string& msg);
int fail(int code, const std::string& msg);

#define ADK_HIP_CHECK(expr)                                                                   \
    do {                                                                                      \
        hipError_t _e = (expr);                                                               \
        if (_e != hipSuccess)                                                                 \
            return ::adk::fail(ADK_ERR_HIP, std::string(#expr) + ": " + hipGetErrorString(_e)); \
    } while (0)

// Flattened arguments of one fused causal conv launch (see adk_causal_conv in the public header).
struct ConvArgs {
    const float* in;   int in_rows, in_ch, in_row0, in_choff, in_gstride;   // in_row0 = (cursor - hist) mod R
    float* out;        int out_rows, out_ch, out_cursor, out_choff;
    const float* res;  int res_rows, res_ch, res_cursor, res_choff, res_gstride;
    const float* w;    const float* wfrag;  const float* bias;   // w: row-major [M][K]; wfrag: MFMA-fragment packed
    int cin_g, cout_g, groups, taps, stride, dilation, up, cout_real;
    int act_in, act_out; float slope;
    int batch, t_out, n_total;      // n_total = batch * t_out GEMM columns
    int ktot;                       // taps * cin_g
    int* err = nullptr;             // sticky device flag word the launch reports to: its program's (adk_program_flags), or nullptr = the device-wide word
    // "Shadow" rings (round 4, split-f16 stream-K kernel only): a second ring of the SAME geometry as `in` / `out` whose every 8-channel
    // group (32 bytes, where the f32 ring holds 8 floats) holds the split-f16 operand form of act(x): [8 x f16 hi][8 x f16 lo], hi = f16(y),
    // lo = f16((y - hi) * 2048), y = act(x) with the READERS' input activation.  Written once by the producer's epilogue (out_sh), it
    // replaces the activation + split a consumer otherwise redoes for every staged element -- once per tap and per 64-row m-tile
    // (11 x 4 times for the 256-channel grouped K11 convs).  Same values, bit for bit.
    const float* in_sh = nullptr;   // shadow of the input ring (nullptr: stage from `in` and convert)
    float* out_sh = nullptr;        // shadow of the output ring, to be written beside `out` (nullptr: none)
    int sh_act = 0; float sh_slope = 0.f;      // activation the output shadow carries
};
inline int* conv_err_word(const ConvArgs& a);

// expm1(x) for x <= 0 (the negative branch of torch.nn.ELU, layers/activation_function.py:18-22 -> torch.nn.ELU).  The library
// expm1f costs ~30 VALU instructions per element, and the ELU convs convert every activation they stage: measured with
// s_memtime stamps (profiles/r2_sk16_timeline.md) the conversion was 2900 of the 4700 cycles of an iteration of the encoder's
// K7 256-channel conv.  Here: |x| < 0.4 -- x + x^2 (1/2 + x/6 + ... + x^5/5040) (truncation < 1.5e-8 relative); else
// exp2(x log2 e) - 1 on the transcendental unit (result in [-1, -0.33]: no cancellation).  Max relative error against fp64
// 1.4e-7 (1.2 ulp; numpy restatement in tests/test_cabi.py), NaN propagates, -inf -> -1.
__device__ __forceinline__ float expm1_neg(float x) {
    const float q = fmaf(fmaf(fmaf(fmaf(fmaf(1.f / 5040.f, x, 1.f / 720.f), x, 1.f / 120.f), x, 1.f / 24.f), x, 1.f / 6.f), x, 0.5f);
    const float p = fmaf(x * x, q, x);
    const float e = __builtin_amdgcn_exp2f(x * 1.44269504088896341f) - 1.0f;
    return x > -0.4f ? p : e;
}

__device__ __forceinline__ float act_apply(float x, int act, float slope) {
    // torch.nn.ELU(alpha=1): x > 0 ? x : expm1(x); LeakyReLU: x > 0 ? x : x*slope; Tanh
    if (act == ADK_ACT_ELU) return x > 0.f ? x : expm1_neg(x);
    if (act == ADK_ACT_LEAKY) return x > 0.f ? x : x * slope;
    if (act == ADK_ACT_TANH) return tanhf(x);
    return x;
}

struct RingMeanArgs {
    const float* src[4]; int src_rows[4], src_cursor[4];
    float* out; int out_rows, out_cursor;
    int n, channels, batch, t;
};
int launch_ring_mean(const RingMeanArgs& m, hipStream_t s);
int launch_hist_replicate(float* ring, int rows, int channels, int cursor, int hist, int batch, hipStream_t s);
int launch_conv_direct(const ConvArgs& a, hipStream_t s);
bool conv_cin1_write_ok(const ConvArgs& a);           // the Cin = 1 K7 conv + the ring write in front of it as one launch (conv_direct.hip)
int launch_conv_cin1_write(const ConvArgs& a, const float* src, hipStream_t s);      // ADK_ERR_STATE: not for this call
// scratch of the stream-K conv: partial accumulators of cut tiles + publish flags (zeroed once at
// allocation; flags carry a per-launch epoch, so they are never reset)
struct Workspace { float* ptr = nullptr; size_t bytes = 0; size_t flags_offset = 0; unsigned epoch = 0;
                   int workgroups = 0; };   // persistent workgroups per stream-K launch; 0 = the whole chip (256 CUs x 2)
int launch_conv_mfma(const ConvArgs& a, hipStream_t s, Workspace& ws);   // needs wfrag, cin_g % 32 == 0
size_t conv_mfma_workspace_bytes(size_t* flags_offset);
bool conv_rl_supported(const ConvArgs& a);          // rows-in-LDS kernel (stride 1, 32/64 channels per group, time-rich)
bool conv_rl_preferred(const ConvArgs& a);          // AUTO heuristic: enough (stream, group, tile) workgroups to fill the chip
int launch_conv_rl(const ConvArgs& a, hipStream_t s);
bool conv_rl16_supported(const ConvArgs& a);        // split-f16 rows-in-LDS kernel (wfrag = adk_pack_weights_split16 layout)
bool conv_rl16_preferred(const ConvArgs& a);
int launch_conv_rl16(const ConvArgs& a, hipStream_t s);
bool conv_rl16_fusable(const ConvArgs& a1, const ConvArgs& a2);      // residual unit (K7 conv -> 1x1 + residual) as one launch
int launch_conv_rl16_fused(const ConvArgs& a1, const ConvArgs& a2, hipStream_t s);   // ADK_ERR_STATE: not fusable for this call
// a whole residual chain (A_0, B_0 + residual, A_1, ...) as one launch, activations resident in LDS (conv_rb16.hip)
bool conv_rb16_fusable(const ConvArgs* c, int n);
int launch_conv_rb16(const ConvArgs* c, int n, const int* keep, hipStream_t s);   // ADK_ERR_STATE: not fusable for this call
const char* conv_rb16_name(const ConvArgs* c, int n);
// conv_out (1x1, 192 -> 64) + activation + the last up-sampler's transposed conv as one streaming launch (conv_ou16.hip)
bool conv_ou16_fusable(const ConvArgs& a1, const ConvArgs& a2);
int launch_conv_ou16(const ConvArgs& a1, const ConvArgs& a2, hipStream_t s);     // ADK_ERR_STATE: not fusable for this call
// the last conv_out (1x1, 96 -> 32) + activation + the output conv (K7, 32 -> 1) + its output activation as one streaming launch (conv_oc16.hip)
bool conv_oc16_fusable(const ConvArgs& a1, const ConvArgs& a2);
int launch_conv_oc16(const ConvArgs& a1, const ConvArgs& a2, hipStream_t s);     // ADK_ERR_STATE: not fusable for this call
bool conv_up16_supported(const ConvArgs& a);        // streaming kernel of the last up-sampling stage (64 -> s*Cout <= 96 rows, 2 taps)
int launch_conv_up16(const ConvArgs& a, hipStream_t s);
int launch_conv_sk16(const ConvArgs& a, hipStream_t s, Workspace& ws);   // split-f16 stream-K (same shapes as launch_conv_mfma)
int conv_sk16_pick(const ConvArgs& a);
bool conv_gv16_preferred(const ConvArgs& a);        // few columns (n_total <= 32): the GEMV-shaped kernel launch_conv_sk16 hands them to
int conv_set_option(const char* name, int value);  // conv_mfma.hip: 0 = set, 1 = not a conv option
int rvq_set_option(const char* name, int value);   // rvq.hip: 0 = set, 1 = not an RVQ option, -1 = bad value
int launch_pack_split16(const float* w, float* out, int groups, int cout_g, int ktot, hipStream_t s);
int* flags_word();                                   // address of the sticky debug/error flags ON THE CURRENT DEVICE
inline int* conv_err_word(const ConvArgs& a) { return a.err ? a.err : flags_word(); }
constexpr int kMaxDevices = 64;
// Per-device pool of sticky flag words (rvq.hip): every program owns one slot for its lifetime.  Since round 6 the words are PINNED HOST memory
// mapped into the device: kernels atomicOr into them over the bus (failure paths only), the host reads and clears with one atomic exchange --
// no kernel, no copy.  `device` must be current.  flag_pool_fetch waits for the stream first; flag_pool_take does not (the caller knows the
// work has finished: an event); flag_pool_fetch_all ORs and clears EVERY slot of the device plus the device-wide word after a device
// synchronisation -- it needs no list of programs and takes no lock programs wait for.
int flag_pool_acquire(int device, int** word);
void flag_pool_release(int device, int* word);
int flag_pool_fetch(int device, int* word, hipStream_t s, int* v);
int flag_pool_take(int device, int* word, int* v);
int flag_pool_fetch_all(int device, int* acc);
inline int current_device() { int d = 0; (void)hipGetDevice(&d); return (d >= 0 && d < kMaxDevices) ? d : 0; }
// Makes `device` current for the lifetime of the object (programs are bound to the device they were created on,
// whatever device the calling thread has current); restores the previous one.
// device that owns a device pointer (the current device when the runtime does not know the pointer)
inline int device_of(const void* ptr) {
    hipPointerAttribute_t at;
    if (ptr && hipPointerGetAttributes(&at, ptr) == hipSuccess && at.type == hipMemoryTypeDevice && at.device >= 0 && at.device < kMaxDevices)
        return at.device;
    (void)hipGetLastError();
    return current_device();
}
struct DeviceGuard {
    int prev = -1;
    explicit DeviceGuard(int device) {
        int cur = -1;
        if (hipGetDevice(&cur) == hipSuccess && cur != device) { if (hipSetDevice(device) == hipSuccess) prev = cur; }
    }
    ~DeviceGuard() { if (prev >= 0) (void)hipSetDevice(prev); }
};
int launch_pack_weights(const float* w, float* out, int groups, int cout_g, int ktot, hipStream_t s);
bool conv_mfma_supported(const ConvArgs& a);
int conv_mfma_pick(const ConvArgs& a);
const char* conv_mfma_cfg_name(int pick);
void conv_mfma_force_cfg(int cfg);
int streamk_plan(long long tiles, int nchunks, int cap, int out[4]);
long long streamk_range_start(long long tiles, int nchunks, const int plan[4], int r);

}  // namespace adk
