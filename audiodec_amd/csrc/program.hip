// Host side of libaudiodec_hip.so: error plumbing, the op-level conv entry point and the "program"
// executor (fixed launch sequence of one model half over B streams with ring-buffer state).
// Replaces the Python module traversal of StreamGenerator.encode/decode
// (models/autoencoder/AudioDec.py:228-247, models/vocoder/HiFiGAN.py:268-296).
#include "adk_common.h"
#include <vector>
#include <cstring>
#include <cstdio>
#include <cstdlib>
#include <mutex>
#include <atomic>

namespace adk {

static thread_local std::string g_err;
void set_error(const std::string& msg) { g_err = msg; }
int fail(int code, const std::string& msg) { g_err = msg; return code; }

static int build_args(const adk_conv_desc& d, const adk_ring_view& in, const adk_ring_view& out,
                      const adk_ring_view& res, int batch, int t_out, ConvArgs& a) {
    if ((!d.w && !d.w_frag) || !in.base || !out.base) return fail(ADK_ERR_ARG, "conv: null weight/in/out pointer");
    if (d.cin_g <= 0 || d.cout_g <= 0 || d.groups <= 0 || d.taps <= 0 || d.stride <= 0 || d.dilation <= 0 ||
        d.up <= 0 || d.cout_real <= 0 || d.hist < 0)
        return fail(ADK_ERR_SHAPE, "conv: non-positive geometry");
    if (batch < 0 || t_out < 0) return fail(ADK_ERR_SHAPE, "conv: negative batch / t_out");
    if ((long long)batch * t_out > 0x7fffffffLL) return fail(ADK_ERR_SHAPE, "conv: batch * t_out overflows");
    const long long M = (long long)d.groups * d.cout_g;
    if (M != (long long)d.up * d.cout_real) return fail(ADK_ERR_SHAPE, "conv: groups*cout_g != up*cout_real");
    if (d.up > 1 && d.groups != 1) return fail(ADK_ERR_SHAPE, "conv: transposed form needs groups == 1");
    // input window [cursor - hist, cursor - hist + (t_out-1)*stride + (taps-1)*dilation] must fit the ring
    const long long span = (long long)(t_out > 0 ? (t_out - 1) : 0) * d.stride + (long long)(d.taps - 1) * d.dilation + 1;
    if (span > in.rows || d.hist >= in.rows + 1) return fail(ADK_ERR_SHAPE, "conv: input window exceeds the ring");
    if (in.cursor < 0 || in.cursor >= in.rows || out.cursor < 0 || out.cursor >= out.rows)
        return fail(ADK_ERR_SHAPE, "conv: cursor outside ring");
    if ((long long)t_out * d.up > out.rows) return fail(ADK_ERR_SHAPE, "conv: output rows exceed the ring");
    if (in.ch_off < 0 || in.ch_off + (long long)(d.groups - 1) * d.in_group_stride + d.cin_g > in.channels)
        return fail(ADK_ERR_SHAPE, "conv: input channels exceed the ring row");
    if (out.ch_off < 0 || out.ch_off + d.cout_real > out.channels)
        return fail(ADK_ERR_SHAPE, "conv: output channels exceed the ring row");
    if (res.base) {
        if (d.up != 1) return fail(ADK_ERR_SHAPE, "conv: residual add needs up == 1");
        if (res.cursor < 0 || res.cursor >= res.rows || t_out > res.rows)
            return fail(ADK_ERR_SHAPE, "conv: residual cursor/rows");
        if (res.ch_off < 0 || res.ch_off + (long long)(d.groups - 1) * d.res_group_stride + d.cout_g > res.channels)
            return fail(ADK_ERR_SHAPE, "conv: residual channels exceed the ring row");
    }
    if (d.act_in < 0 || d.act_in > ADK_ACT_TANH || d.act_out < 0 || d.act_out > ADK_ACT_TANH)
        return fail(ADK_ERR_ARG, "conv: unknown activation");
    a.in = in.base; a.in_rows = in.rows; a.in_ch = in.channels; a.in_choff = in.ch_off; a.in_gstride = d.in_group_stride;
    a.in_row0 = ((in.cursor - d.hist) % in.rows + in.rows) % in.rows;
    a.out = out.base; a.out_rows = out.rows; a.out_ch = out.channels; a.out_cursor = out.cursor; a.out_choff = out.ch_off;
    a.res = res.base; a.res_rows = res.rows; a.res_ch = res.channels; a.res_cursor = res.cursor; a.res_choff = res.ch_off;
    a.res_gstride = d.res_group_stride;
    a.w = d.w; a.wfrag = d.w_frag; a.bias = d.bias;
    a.cin_g = d.cin_g; a.cout_g = d.cout_g; a.groups = d.groups; a.taps = d.taps; a.stride = d.stride;
    a.dilation = d.dilation; a.up = d.up; a.cout_real = d.cout_real;
    a.act_in = d.act_in; a.act_out = d.act_out; a.slope = d.act_in_slope;
    a.batch = batch; a.t_out = t_out; a.n_total = batch * t_out; a.ktot = d.taps * d.cin_g;
    return ADK_OK;
}


static int ensure_workspace(Workspace& w) {
    size_t flags_offset = 0;
    const size_t need = conv_mfma_workspace_bytes(&flags_offset);
    if (w.ptr && w.bytes >= need) return ADK_OK;
    if (w.ptr) (void)hipFree(w.ptr);
    w.ptr = nullptr; w.bytes = 0;
    ADK_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&w.ptr), need));
    ADK_HIP_CHECK(hipMemset(w.ptr, 0, need));         // publish flags start below every epoch
    w.bytes = need; w.flags_offset = flags_offset; w.epoch = 0;
    return ADK_OK;
}

// run-time options: read once from the environment (read_env), set by adk_set_option from any host thread -- atomics, like the rvq and conv options
static std::atomic<int> g_use_rl{-1};       // ADK_CONV_RL=0 disables the rows-in-LDS kernel in AUTO mode (tuning aid)
static std::atomic<int> g_use_up{-1};       // ADK_CONV_UP16=0 disables the up-sampling streamer in AUTO mode (tuning aid)
static std::atomic<int> g_use_chain{-1};    // ADK_CHAIN=0: residual chains run op by op (A/B against the per-op kernels)
static std::atomic<int> g_chain_max_c{-1};  // adk_set_option("chain_max_channels") / ADK_CHAIN_MAXC
static std::atomic<int> g_chain_min_c{-1};  // adk_set_option("chain_min_channels") / ADK_CHAIN_MINC
static std::atomic<int> g_chain_min_blocks{-1};   // adk_set_option("chain_min_blocks") / ADK_CHAIN_MIN_BLOCKS: fewer (stream, group) workgroups -> per-op launches
static std::atomic<int> g_use_ou{-1};       // adk_set_option("conv_ou16") / ADK_CONV_OU16=0: conv_out and the last up-sampler stay two launches (A/B, tests)
static std::atomic<int> g_use_oc{-1};       // adk_set_option("conv_oc16") / ADK_CONV_OC16=0: the last conv_out and the output conv stay two launches
static std::atomic<int> g_use_cw{-1};       // adk_set_option("conv_cin1w") / ADK_CONV_CIN1W=0: the encoder's ring write stays a launch of its own

static bool is_split16(int impl) {
    return impl == ADK_IMPL_SPLIT16 || impl == ADK_IMPL_SPLIT16_ROWS || impl == ADK_IMPL_SPLIT16_SK || impl == ADK_IMPL_SPLIT16_UP;
}
static void read_env() {
    if (g_use_rl < 0) { const char* e = getenv("ADK_CONV_RL"); g_use_rl = e ? atoi(e) : 1; }
    if (g_use_up < 0) { const char* e = getenv("ADK_CONV_UP16"); g_use_up = e ? atoi(e) : 1; }
    if (g_use_chain < 0) { const char* e = getenv("ADK_CHAIN"); g_use_chain = e ? atoi(e) : 1; }
    if (g_chain_max_c < 0) { const char* e = getenv("ADK_CHAIN_MAXC"); g_chain_max_c = e ? atoi(e) : 128; }
    if (g_chain_min_c < 0) { const char* e = getenv("ADK_CHAIN_MINC"); g_chain_min_c = e ? atoi(e) : 0; }
    // measured crossover (tools/chain_crossover.py, profiles/r3_chain_crossover.log): below ~160 (stream, group) pairs the per-op launches,
    // which spread one stream's time tiles over many CUs, are faster than one workgroup per pair walking the whole chain
    if (g_chain_min_blocks < 0) { const char* e = getenv("ADK_CHAIN_MIN_BLOCKS"); g_chain_min_blocks = e ? atoi(e) : 0; }
    if (g_use_ou < 0) { const char* e = getenv("ADK_CONV_OU16"); g_use_ou = e ? atoi(e) : 1; }
    if (g_use_oc < 0) { const char* e = getenv("ADK_CONV_OC16"); g_use_oc = e ? atoi(e) : 1; }
    if (g_use_cw < 0) { const char* e = getenv("ADK_CONV_CIN1W"); g_use_cw = e ? atoi(e) : 1; }
}

static int run_conv(const ConvArgs& a, int impl, hipStream_t s, Workspace& ws) {
    read_env();
    const bool ok = conv_mfma_supported(a);
    if (impl == ADK_IMPL_MFMA_ROWS) {
        if (!conv_rl_supported(a)) return fail(ADK_ERR_SHAPE, "conv: rows-in-LDS kernel needs stride 1, 32/64 channels per group, w_frag");
        return launch_conv_rl(a, s);
    }
    if (a.out_sh && impl != ADK_IMPL_SPLIT16_SK)
        return fail(ADK_ERR_STATE, "conv: an op that writes a shadow ring must run on the split-f16 stream-K kernel (impl = ADK_IMPL_SPLIT16_SK)");
    if (is_split16(impl)) {
        // split-f16 kernels (w_frag in the adk_pack_weights_split16 layout): the up-sampling streamer for its one layer shape,
        // rows-in-LDS when it fills the chip, else stream-K
        if (impl == ADK_IMPL_SPLIT16_UP && !conv_up16_supported(a))
            return fail(ADK_ERR_SHAPE, "conv: the up-sampling streamer takes 2-tap transposed convs with 64 input channels and <= 96 GEMM rows");
        if (impl == ADK_IMPL_SPLIT16_UP || (impl == ADK_IMPL_SPLIT16 && g_use_up && conv_up16_supported(a)))
            return launch_conv_up16(a, s);
        if (impl == ADK_IMPL_SPLIT16_ROWS && !conv_rl16_supported(a))
            return fail(ADK_ERR_SHAPE, "conv: split-f16 rows-in-LDS kernel needs stride 1, 32/64 channels per group, K in {3,7,11}, split16 w_frag");
        if (impl == ADK_IMPL_SPLIT16_ROWS || (impl == ADK_IMPL_SPLIT16 && g_use_rl && conv_rl16_preferred(a)))
            return launch_conv_rl16(a, s);
        if (!ok) return fail(ADK_ERR_SHAPE, "conv: split-f16 kernel needs w_frag, cin_g % 32 == 0 and 16-byte aligned rows");
        int rc = ensure_workspace(ws);
        if (rc != ADK_OK) return rc;
        return launch_conv_sk16(a, s, ws);
    }
    const bool want_mfma = (impl == ADK_IMPL_MFMA) || (impl == ADK_IMPL_AUTO && ok && a.groups * a.cout_g >= 32);
    if (impl == ADK_IMPL_MFMA && !ok)
        return fail(ADK_ERR_SHAPE, "conv: MFMA kernel needs w_frag, cin_g % 32 == 0 and 16-byte aligned rows");
    if (want_mfma) {
        if (impl == ADK_IMPL_AUTO && g_use_rl && conv_rl_preferred(a)) return launch_conv_rl(a, s);
        int rc = ensure_workspace(ws);
        if (rc != ADK_OK) return rc;
        return launch_conv_mfma(a, s, ws);
    }
    // VALU kernel: Cin = 1 / Cout = 1 layers and anything the matrix-core kernel does not take
    if (!a.w) return fail(ADK_ERR_ARG, "conv: the VALU kernel needs row-major weights (w)");
    return launch_conv_direct(a, s);
}

// name of the kernel run_conv would launch for these arguments (profiles, tests)
static std::string conv_kernel_name(const ConvArgs& a, int impl) {
    read_env();
    if (is_split16(impl)) {
        if (impl == ADK_IMPL_SPLIT16_UP || (impl == ADK_IMPL_SPLIT16 && g_use_up && conv_up16_supported(a))) return "conv_up16<64>";
        const bool rows = impl == ADK_IMPL_SPLIT16_ROWS || (impl == ADK_IMPL_SPLIT16 && g_use_rl && conv_rl16_preferred(a));
        if (rows) return a.cin_g == 32 ? "conv_rl16<32>" : "conv_rl16<64>";
        if (conv_gv16_preferred(a)) return "conv_gv16<32>";
        return std::string(conv_mfma_cfg_name(conv_sk16_pick(a))).replace(0, 7, "conv_sk16");
    }
    const bool mf = impl != ADK_IMPL_DIRECT && conv_mfma_supported(a) && (impl == ADK_IMPL_MFMA || impl == ADK_IMPL_MFMA_ROWS || a.groups * a.cout_g >= 32);
    const bool rl = mf && ((impl == ADK_IMPL_MFMA_ROWS && conv_rl_supported(a)) || (impl == ADK_IMPL_AUTO && g_use_rl && conv_rl_preferred(a)));
    if (rl) return a.cin_g == 32 ? "conv_rl<32>" : "conv_rl<64>";
    if (mf) return conv_mfma_cfg_name(conv_mfma_pick(a));
    return a.groups * a.cout_g == 1 ? "conv_cout1" : (a.cin_g == 1 && a.taps == 7 ? "conv_cin1" : "conv_direct");
}

}  // namespace adk

using namespace adk;

extern "C" const char* adk_last_error(void) { return g_err.c_str(); }
extern "C" int adk_abi_version(void) { return ADK_ABI_VERSION; }
extern "C" int adk_set_conv_cfg(int32_t cfg) { conv_mfma_force_cfg(cfg); return ADK_OK; }
extern "C" int adk_set_option(const char* name, int32_t value) {
    if (!name) return fail(ADK_ERR_ARG, "adk_set_option: null name");
    read_env();
    if (!strcmp(name, "chain_max_channels")) { g_chain_max_c = value < 0 ? 0 : value; return ADK_OK; }
    if (!strcmp(name, "chain_min_channels")) { g_chain_min_c = value < 0 ? 0 : value; return ADK_OK; }
    if (!strcmp(name, "chain_min_blocks")) { g_chain_min_blocks = value < 0 ? 0 : value; return ADK_OK; }
    if (!strcmp(name, "conv_ou16")) { g_use_ou = value != 0; return ADK_OK; }
    if (!strcmp(name, "conv_oc16")) { g_use_oc = value != 0; return ADK_OK; }
    if (!strcmp(name, "conv_cin1w")) { g_use_cw = value != 0; return ADK_OK; }
    if (conv_set_option(name, value) == 0) return ADK_OK;
    {
        const int r = rvq_set_option(name, value);
        if (r == 0) return ADK_OK;
        if (r < 0) return fail(ADK_ERR_ARG, std::string("adk_set_option: bad value for ") + name);
    }
    return fail(ADK_ERR_ARG, std::string("adk_set_option: unknown option ") + name);
}

extern "C" int adk_streamk_plan(int64_t tiles, int32_t chunks, int32_t cap, int32_t* plan) {
    if (!plan || tiles < 1 || tiles >= (1ll << 31) / 64 || chunks < 1 || chunks > (1 << 20) || cap < 0) return fail(ADK_ERR_ARG, "adk_streamk_plan: bad argument");
    int out[4];
    const int rc = streamk_plan(tiles, chunks, cap, out);
    for (int i = 0; i < 4; ++i) plan[i] = out[i];
    return rc;
}

extern "C" int64_t adk_streamk_range_start(int64_t tiles, int32_t chunks, const int32_t* plan, int32_t r) {
    if (!plan || tiles < 1 || chunks < 1 || plan[0] < 1 || r < 0 || r > plan[0]) return -1;
    const int p[4] = {plan[0], plan[1], plan[2], plan[3]};
    return streamk_range_start(tiles, chunks, p, r);
}

extern "C" int adk_causal_conv(const adk_conv_desc* d, adk_ring_view in, adk_ring_view out, adk_ring_view res,
                               int32_t batch, int32_t t_out, int32_t impl, void* stream) {
    if (!d) return fail(ADK_ERR_ARG, "adk_causal_conv: null descriptor");
    ConvArgs a;
    int rc = build_args(*d, in, out, res, batch, t_out, a);
    if (rc != ADK_OK) return rc;
    DeviceGuard guard(device_of(out.base));
    static thread_local Workspace tls_ws[kMaxDevices];   // op-level calls: one scratch per host thread and device
    return run_conv(a, impl, static_cast<hipStream_t>(stream), tls_ws[current_device()]);
}

extern "C" int adk_causal_conv_describe(const adk_conv_desc* d, adk_ring_view in, adk_ring_view out, adk_ring_view res,
                                        int32_t batch, int32_t t_out, int32_t impl, char* buf, int32_t n) {
    if (!d || !buf || n <= 0) return fail(ADK_ERR_ARG, "adk_causal_conv_describe: null argument");
    ConvArgs a;
    int rc = build_args(*d, in, out, res, batch, t_out, a);
    if (rc != ADK_OK) return rc;
    snprintf(buf, n, "%s", conv_kernel_name(a, impl).c_str());
    return ADK_OK;
}

extern "C" int adk_causal_conv_time(const adk_conv_desc* d, adk_ring_view in, adk_ring_view out, adk_ring_view res,
                                    int32_t batch, int32_t t_out, int32_t impl, int32_t iters, void* stream, float* avg_us) {
    if (!d || !avg_us || iters <= 0) return fail(ADK_ERR_ARG, "adk_causal_conv_time: bad arguments");
    ConvArgs a;
    int rc = build_args(*d, in, out, res, batch, t_out, a);
    if (rc != ADK_OK) return rc;
    DeviceGuard guard(device_of(out.base));
    hipStream_t s = static_cast<hipStream_t>(stream);
    static thread_local Workspace tls_ws[kMaxDevices];
    Workspace& ws = tls_ws[current_device()];
    for (int i = 0; i < 5 && rc == ADK_OK; ++i) rc = run_conv(a, impl, s, ws);       // warm-up (first-use attributes, caches)
    if (rc != ADK_OK) return rc;
    hipEvent_t e0, e1;
    ADK_HIP_CHECK(hipEventCreate(&e0)); ADK_HIP_CHECK(hipEventCreate(&e1));
    ADK_HIP_CHECK(hipEventRecord(e0, s));
    for (int i = 0; i < iters && rc == ADK_OK; ++i) rc = run_conv(a, impl, s, ws);
    ADK_HIP_CHECK(hipEventRecord(e1, s));
    ADK_HIP_CHECK(hipEventSynchronize(e1));
    float ms = 0.f;
    ADK_HIP_CHECK(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    *avg_us = 1e3f * ms / (float)iters;
    return rc;
}

extern "C" int64_t adk_packed_weight_floats(int32_t groups, int32_t cout_g, int32_t ktot) {
    if (groups <= 0 || cout_g <= 0 || ktot <= 0 || ktot % 8) return -1;
    return (int64_t)groups * ((cout_g + 31) / 32) * ((ktot + 63) / 64 * 8) * 256;
}

extern "C" int adk_pack_weights_mfma(const float* w, float* out, int32_t groups, int32_t cout_g, int32_t ktot, void* stream) {
    if (!w || !out) return fail(ADK_ERR_ARG, "adk_pack_weights_mfma: null pointer");
    if (groups <= 0 || cout_g <= 0 || ktot <= 0 || ktot % 8) return fail(ADK_ERR_SHAPE, "adk_pack_weights_mfma: need ktot % 8 == 0");
    DeviceGuard guard(device_of(out));
    return launch_pack_weights(w, out, groups, cout_g, ktot, static_cast<hipStream_t>(stream));
}

extern "C" int64_t adk_packed_weight_floats_split16(int32_t groups, int32_t cout_g, int32_t ktot) {
    if (groups <= 0 || cout_g <= 0 || ktot <= 0 || ktot % 16) return -1;
    return (int64_t)groups * ((cout_g + 31) / 32) * ((ktot + 63) / 64 * 4) * 512;
}

extern "C" int adk_pack_weights_split16(const float* w, float* out, int32_t groups, int32_t cout_g, int32_t ktot, void* stream) {
    if (!w || !out) return fail(ADK_ERR_ARG, "adk_pack_weights_split16: null pointer");
    if (groups <= 0 || cout_g <= 0 || ktot <= 0 || ktot % 16) return fail(ADK_ERR_SHAPE, "adk_pack_weights_split16: need ktot % 16 == 0");
    DeviceGuard guard(device_of(out));
    return launch_pack_split16(w, out, groups, cout_g, ktot, static_cast<hipStream_t>(stream));
}

// ------------------------------------------------------------------------------------------------
struct adk_program {
    std::vector<adk_op_desc> ops;
    std::vector<adk_ring_desc> rings;
    std::vector<int32_t> rows;      // ring length (arena rings)
    std::vector<int32_t> cursor;
    int batch = 0, max_frames = 0, n_ext = 0;
    int device = 0;                 // HIP device the program lives on (the owner of its arena)
    const float* weights = nullptr; int64_t weights_floats = 0;
    float* arena = nullptr; int64_t arena_floats = 0;
    Workspace ws;
    int* flags = nullptr;           // this program's sticky device flag word (bits as adk_debug_flags): what its launches report to; a slot of the device's pool
    bool profiling = false;
    bool fresh = true;              // no step since create / reset (ADK_OP_HIST_REPLICATE runs only then)
    long long steps_since_reset = 0;   // steps taken since create / reset / set_fresh(1), rewinds subtracted: adk_program_rewind, applied any
                                       // number of times in a row (ring extra_rows), restores `fresh` exactly when it is back at the first step
    bool replaying = false;         // the step in progress was asked for with ADK_STEP_REPLAY
    std::vector<hipEvent_t> ev;     // n_ops + 1 events when profiling
    std::vector<float> last_ms;
    // HIP-graph replay of the steady state (adk_program_set_graph): one captured graph per cursor phase
    bool graph = false;
    int period = 0;                 // cursor states of full-size steps repeat with this period (0: no usable period)
    int g_lo = 0, g_hi = 0;         // ops [g_lo, g_hi) are captured; the ops touching caller buffers at either end stay eager
    std::vector<char> seen;         // phase was run eagerly once (kernel attributes, symbol addresses are set up)
    std::vector<hipGraphExec_t> gexec;
    long long replays = 0, captures = 0;
    // deferred flag checks (adk_program_flags_post / _poll): events of the last ADK_POST_SLOTS posts (the flag word itself is pinned host memory)
    hipEvent_t post_ev[ADK_POST_SLOTS] = {};
    long long post_next = 0;
};

static int program_fetch_clear(adk_program* p, hipStream_t s, int* v) { return flag_pool_fetch(p->device, p->flags, s, v); }

extern "C" int adk_program_create(const adk_op_desc* ops, int32_t n_ops, const adk_ring_desc* rings, int32_t n_rings,
                                  int32_t batch, int32_t max_frames, const float* weights, int64_t weights_floats,
                                  float* arena, int64_t arena_floats, adk_program** out) {
    if (!ops || !rings || !out || n_ops <= 0 || n_rings <= 0) return fail(ADK_ERR_ARG, "program_create: null/empty input");
    if (batch <= 0 || max_frames <= 0) return fail(ADK_ERR_SHAPE, "program_create: batch and max_frames must be positive");
    if (!weights || !arena) return fail(ADK_ERR_ARG, "program_create: null weights/arena");
    adk_program* p = new adk_program();
    p->ops.assign(ops, ops + n_ops);
    p->rings.assign(rings, rings + n_rings);
    p->rows.resize(n_rings); p->cursor.assign(n_rings, 0);
    p->batch = batch; p->max_frames = max_frames;
    p->device = current_device();
    {   // a program lives on the device that owns its arena, whatever device the calling thread has current
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, arena) == hipSuccess && at.type == hipMemoryTypeDevice && at.device >= 0 && at.device < kMaxDevices)
            p->device = at.device;
        else
            (void)hipGetLastError();          // not a device pointer known to the runtime: keep the current device
    }
    DeviceGuard guard(p->device);
    p->weights = weights; p->weights_floats = weights_floats; p->arena = arena; p->arena_floats = arena_floats;
    auto bail = [&](int code, const std::string& m) { delete p; return fail(code, m); };
    for (int i = 0; i < n_rings; ++i) {
        const adk_ring_desc& r = rings[i];
        if (r.channels <= 0 || r.hist < 0 || r.rate <= 0) return bail(ADK_ERR_SHAPE, "program_create: bad ring desc");
        if (r.external >= 0) {
            if (r.hist != 0 || r.extra_rows != 0) return bail(ADK_ERR_SHAPE, "program_create: external rings carry no history");
            p->rows[i] = 0;
            if (r.external + 1 > p->n_ext) p->n_ext = r.external + 1;
        } else {
            if (r.extra_rows < 0) return bail(ADK_ERR_SHAPE, "program_create: negative extra_rows");
            const long long rows = (long long)r.hist + (long long)max_frames * r.rate + (long long)r.extra_rows;
            if (rows > 0x7fffffffLL) return bail(ADK_ERR_SHAPE, "program_create: ring too long");
            p->rows[i] = (int32_t)rows;
            const long long need = r.arena_off + (long long)batch * rows * r.channels;
            if (r.arena_off < 0 || r.arena_off % 4 || need > arena_floats)
                return bail(ADK_ERR_SHAPE, "program_create: ring does not fit the arena (or offset not 16-byte aligned)");
        }
    }
    for (int i = 0; i < n_ops; ++i) {
        const adk_op_desc& o = ops[i];
        auto ring_ok = [&](int id) { return id >= 0 && id < n_rings; };
        if (o.kind == ADK_OP_CONV) {
            if (!ring_ok(o.in_ring) || !ring_ok(o.out_ring) || (o.res_ring >= 0 && !ring_ok(o.res_ring)))
                return bail(ADK_ERR_ARG, "program_create: op references an unknown ring");
            const long long wn = (long long)o.conv.groups * o.conv.cout_g * o.conv.taps * o.conv.cin_g;
            if (o.w_off < 0 && o.wf_off < 0) return bail(ADK_ERR_ARG, "program_create: op has no weights");
            if (o.w_off >= 0 && (o.w_off % 4 || o.w_off + wn > weights_floats)) return bail(ADK_ERR_SHAPE, "program_create: weight offset out of range");
            if (o.wf_off >= 0) {
                const bool s16 = is_split16(o.impl);
                const long long wfn = s16 ? adk_packed_weight_floats_split16(o.conv.groups, o.conv.cout_g, o.conv.taps * o.conv.cin_g)
                                          : adk_packed_weight_floats(o.conv.groups, o.conv.cout_g, o.conv.taps * o.conv.cin_g);
                if (wfn < 0 || o.wf_off % 4 || o.wf_off + wfn > weights_floats) return bail(ADK_ERR_SHAPE, "program_create: packed weight offset out of range");
            }
            if (o.b_off >= 0 && (o.b_off % 4 || o.b_off + (long long)o.conv.groups * o.conv.cout_g > weights_floats))
                return bail(ADK_ERR_SHAPE, "program_create: bias offset out of range");
            if (o.rate_out <= 0) return bail(ADK_ERR_SHAPE, "program_create: rate_out must be positive");
            for (int side = 0; side < 2; ++side) {
                const int sh = (side ? o.out_shadow : o.in_shadow) - 1, of = side ? o.out_ring : o.in_ring;
                if (sh < 0) continue;
                if (!ring_ok(sh) || sh == of || rings[sh].external >= 0 || rings[of].external >= 0)
                    return bail(ADK_ERR_ARG, "program_create: a shadow ring must be an arena ring of its own, shadowing an arena ring");
                if (rings[sh].channels != rings[of].channels || rings[sh].hist != rings[of].hist || rings[sh].rate != rings[of].rate ||
                    rings[sh].extra_rows != rings[of].extra_rows)
                    return bail(ADK_ERR_SHAPE, "program_create: a shadow ring must have the geometry of its ring");
                // the kernels store [8 x f16 hi][8 x f16 lo] per 8-channel group: half a group at the end of a row would land in the next row
                if (rings[sh].channels % 8 || (side ? o.out_ch_off : o.in_ch_off) % 8)
                    return bail(ADK_ERR_SHAPE, "program_create: a shadowed ring needs channels % 8 == 0 and channel offsets % 8 == 0");
                if (!is_split16(o.impl)) return bail(ADK_ERR_ARG, "program_create: shadow rings are for ADK_IMPL_SPLIT16* ops");
                if (side && o.impl != ADK_IMPL_SPLIT16_SK) return bail(ADK_ERR_ARG, "program_create: an op that writes a shadow ring needs impl = ADK_IMPL_SPLIT16_SK");
                if (side && (o.out_ch_off != 0 || o.conv.cout_real != rings[of].channels))
                    return bail(ADK_ERR_SHAPE, "program_create: an op that writes a shadow ring must write whole rows");
                if (side && (o.shadow_act < 0 || o.shadow_act > ADK_ACT_LEAKY)) return bail(ADK_ERR_ARG, "program_create: shadow_act must be NONE, ELU or LeakyReLU");
            }
            if (o.conv.hist > rings[o.in_ring].hist) return bail(ADK_ERR_SHAPE, "program_create: op needs more history than its ring keeps");
        } else if (o.kind == ADK_OP_RING_WRITE) {
            if (!ring_ok(o.out_ring) || o.ext_src < 0) return bail(ADK_ERR_ARG, "program_create: ring_write needs out_ring and ext_src");
            if (o.ext_src + 1 > p->n_ext) p->n_ext = o.ext_src + 1;
            if ((o.mean_off >= 0) != (o.scale_off >= 0)) return bail(ADK_ERR_ARG, "program_create: mean and scale go together");
        } else if (o.kind == ADK_OP_MEAN) {
            if (!ring_ok(o.out_ring) || o.n_mean < 1 || o.n_mean > 4) return bail(ADK_ERR_ARG, "program_create: mean needs out_ring and 1..4 sources");
            for (int k = 0; k < o.n_mean; ++k) {
                if (!ring_ok(o.mean_rings[k])) return bail(ADK_ERR_ARG, "program_create: mean references an unknown ring");
                if (rings[o.mean_rings[k]].channels != rings[o.out_ring].channels || rings[o.mean_rings[k]].rate != rings[o.out_ring].rate)
                    return bail(ADK_ERR_SHAPE, "program_create: mean sources must match the output ring");
            }
        } else if (o.kind == ADK_OP_HIST_REPLICATE) {
            if (!ring_ok(o.in_ring) || rings[o.in_ring].external >= 0)
                return bail(ADK_ERR_ARG, "program_create: hist_replicate needs an arena ring");
        } else {
            return bail(ADK_ERR_ARG, "program_create: unknown op kind");
        }
    }
    // shadow consistency across ops: every conv that writes a shadowed ring writes the shadow too, with the activation its readers expect
    for (int i = 0; i < n_ops; ++i) {
        const adk_op_desc& rd = ops[i];
        if (rd.kind != ADK_OP_CONV || rd.in_shadow <= 0) continue;
        for (int j = 0; j < n_ops; ++j) {
            const adk_op_desc& wr = ops[j];
            if (wr.kind == ADK_OP_HIST_REPLICATE) continue;
            if (wr.out_ring != rd.in_ring) continue;
            if (wr.kind != ADK_OP_CONV || wr.out_shadow != rd.in_shadow)
                return bail(ADK_ERR_ARG, "program_create: every op that writes a shadowed ring must be a conv carrying the same out_shadow");
            if (wr.shadow_act != rd.conv.act_in || (wr.shadow_act == ADK_ACT_LEAKY && wr.shadow_slope != rd.conv.act_in_slope))
                return bail(ADK_ERR_ARG, "program_create: shadow_act / shadow_slope of a writer differ from the act_in / slope of a reader of the shadow");
        }
    }
    {
        int rc = ensure_workspace(p->ws);
        if (rc != ADK_OK) { delete p; return rc; }
        rc = flag_pool_acquire(p->device, &p->flags);
        if (rc != ADK_OK) {
            if (p->ws.ptr) (void)hipFree(p->ws.ptr);
            delete p;
            return rc;
        }
    }
    *out = p;
    return ADK_OK;
}

extern "C" void adk_program_destroy(adk_program* p) {
    if (!p) return;
    DeviceGuard guard(p->device);
    for (hipGraphExec_t g : p->gexec) if (g) (void)hipGraphExecDestroy(g);
    if (p->ws.ptr) (void)hipFree(p->ws.ptr);
    flag_pool_release(p->device, p->flags);
    for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
    for (hipEvent_t e : p->post_ev) if (e) (void)hipEventDestroy(e);
    delete p;
}

static adk_ring_view view_of(const adk_program* p, int id, int frames, void* const* ext, int ch_off) {
    adk_ring_view v;
    const adk_ring_desc& r = p->rings[id];
    v.channels = r.channels; v.ch_off = ch_off;
    if (r.external >= 0) {
        v.base = static_cast<float*>(ext[r.external]);
        v.rows = frames * r.rate; v.cursor = 0;
    } else {
        v.base = p->arena + r.arena_off; v.rows = p->rows[id]; v.cursor = p->cursor[id];
    }
    return v;
}

static int op_conv_args(adk_program* p, int i, int frames, void* const* ext, ConvArgs& a) {
    const adk_op_desc& o = p->ops[i];
    adk_conv_desc d = o.conv;
    d.w = o.w_off >= 0 ? p->weights + o.w_off : nullptr;
    d.w_frag = o.wf_off >= 0 ? p->weights + o.wf_off : nullptr;
    d.bias = o.b_off >= 0 ? p->weights + o.b_off : nullptr;
    adk_ring_view in = view_of(p, o.in_ring, frames, ext, o.in_ch_off);
    adk_ring_view out = view_of(p, o.out_ring, frames, ext, o.out_ch_off);
    adk_ring_view res; memset(&res, 0, sizeof(res));
    if (o.res_ring >= 0) res = view_of(p, o.res_ring, frames, ext, o.res_ch_off);
    const int rc = build_args(d, in, out, res, p->batch, frames * o.rate_out, a);
    a.err = p->flags;
    if (rc != ADK_OK) return rc;
    // shadow rings (adk_op_desc.in_shadow / out_shadow; geometry checked against their rings at create time): they move with their rings
    if (o.in_shadow > 0) a.in_sh = view_of(p, o.in_shadow - 1, frames, ext, 0).base;
    if (o.out_shadow > 0) { a.out_sh = view_of(p, o.out_shadow - 1, frames, ext, 0).base; a.sh_act = o.shadow_act; a.sh_slope = o.shadow_slope; }
    return ADK_OK;
}

// Can ops i, i+1 run as one launch for a `frames`-hop step?  0: no; 1: a residual unit (conv -> 1x1 + residual: conv_rl16 FUSE);
// 2: the 1x1 conv_out of a vocoder stage + the next stage's activation and transposed conv (conv_ou16);
// 3: the last 1x1 conv_out + activation + the output conv and its activation (conv_oc16)
static int op_pair_kind(adk_program* p, int i, int frames, void* const* ext, ConvArgs& a1, ConvArgs& a2) {
    read_env();
    if (i + 1 >= (int)p->ops.size()) return 0;
    const adk_op_desc &o1 = p->ops[i], &o2 = p->ops[i + 1];
    if (o1.kind != ADK_OP_CONV || o2.kind != ADK_OP_CONV || !o1.fuse_next || o1.impl != ADK_IMPL_SPLIT16) return 0;
    if (o2.in_ring != o1.out_ring || p->rings[o1.out_ring].external >= 0) return 0;
    if (o1.out_shadow > 0 || o2.out_shadow > 0) return 0;        // (only the stream-K kernel's epilogue writes shadow rings)
    if (op_conv_args(p, i, frames, ext, a1) != ADK_OK || op_conv_args(p, i + 1, frames, ext, a2) != ADK_OK) return 0;
    if (o2.impl == ADK_IMPL_AUTO && o2.conv.groups * o2.conv.cout_g == 1)      // (a one-channel conv has no matrix-core form: exact f32 in either arithmetic)
        return g_use_oc && conv_oc16_fusable(a1, a2) ? 3 : 0;
    if (o2.impl != ADK_IMPL_SPLIT16) return 0;
    if (g_use_rl && conv_rl16_fusable(a1, a2)) return 1;
    if (g_use_ou && g_use_up && conv_ou16_fusable(a1, a2)) return 2;
    return 0;
}
static bool op_pair_fusable(adk_program* p, int i, int frames, void* const* ext, ConvArgs& a1, ConvArgs& a2) {
    return op_pair_kind(p, i, frames, ext, a1, a2) == 1;
}

// Can ops [i, i + n) (a residual chain, adk_op_desc.chain) run as one launch for a `frames`-hop step?  Fills c[] / keep[].
constexpr int kMaxChain = 8;
static bool op_chain_fusable(adk_program* p, int i, int frames, void* const* ext, ConvArgs* c, int* keep) {
    read_env();
    const int n = p->ops[i].chain;
    if (!g_use_chain || !g_use_rl || n < 2 || n > kMaxChain || i + n > (int)p->ops.size()) return false;
    if (p->ops[i].conv.cin_g > g_chain_max_c || p->ops[i].conv.cin_g < g_chain_min_c) return false;
    if ((long long)p->batch * p->ops[i].conv.groups < g_chain_min_blocks) return false;
    for (int k = 0; k < n; ++k) {
        const adk_op_desc& o = p->ops[i + k];
        if (o.kind != ADK_OP_CONV || o.impl != ADK_IMPL_SPLIT16) return false;      // (ops that write a shadow ring carry ADK_IMPL_SPLIT16_SK: never fused)
        if (k > 0 && o.chain > 1) return false;
        if (k + 1 < n && (p->rings[o.out_ring].external >= 0 || p->ops[i + k + 1].in_ring != o.out_ring)) return false;
        if (op_conv_args(p, i + k, frames, ext, c[k]) != ADK_OK) return false;
        keep[k] = p->rings[o.out_ring].hist;
    }
    return conv_rb16_fusable(c, n);
}

// Op i writes the caller's rows into a one-channel ring and op i + 1 is the Cin = 1 conv that reads it: one launch (conv_cin1w_kernel)?
static bool op_write_conv_fusable(adk_program* p, int i, int frames, void* const* ext, ConvArgs& a) {
    read_env();
    if (!g_use_cw || i + 1 >= (int)p->ops.size()) return false;
    const adk_op_desc &o1 = p->ops[i], &o2 = p->ops[i + 1];
    if (o1.kind != ADK_OP_RING_WRITE || o2.kind != ADK_OP_CONV || o1.mean_off >= 0 || o1.scale_off >= 0 || o2.in_ring != o1.out_ring) return false;
    if (o2.impl != ADK_IMPL_AUTO || p->rings[o1.out_ring].channels != 1 || p->rings[o1.out_ring].external >= 0 || o2.in_shadow > 0 || o2.out_shadow > 0) return false;
    if (op_conv_args(p, i + 1, frames, ext, a) != ADK_OK) return false;
    return a.t_out == frames * p->rings[o1.out_ring].rate && conv_cin1_write_ok(a);
}

// one op of the launch sequence on stream s; *consumed = how many ops of the sequence this launch covered (a residual unit or a
// whole residual chain run as one kernel), at most max_consume
static int run_op(adk_program* p, int i, int frames, void* const* ext, hipStream_t s, int max_consume = 1, int* consumed = nullptr) {
    const adk_op_desc& o = p->ops[i];
    int rc = ADK_OK;
    if (consumed) *consumed = 1;
    if (o.kind == ADK_OP_RING_WRITE && p->replaying) return ADK_OK;      // ADK_STEP_REPLAY: the rows are still where the rewound step put them
    if (o.kind == ADK_OP_CONV) {
        ConvArgs a, a2;
        if (consumed && o.chain >= 2 && o.chain <= max_consume) {
            ConvArgs c[kMaxChain]; int keep[kMaxChain];
            if (op_chain_fusable(p, i, frames, ext, c, keep)) {
                rc = launch_conv_rb16(c, o.chain, keep, s);
                if (rc == ADK_OK) { *consumed = o.chain; return ADK_OK; }
                if (rc != ADK_ERR_STATE) { g_err = "op " + std::to_string(i) + " (chain): " + g_err; return rc; }
            }
        }
        if (consumed && max_consume >= 2 && o.fuse_next) {
            const int kind = op_pair_kind(p, i, frames, ext, a, a2);
            if (kind) {
                rc = kind == 1 ? launch_conv_rl16_fused(a, a2, s) : (kind == 2 ? launch_conv_ou16(a, a2, s) : launch_conv_oc16(a, a2, s));
                if (rc == ADK_OK) { *consumed = 2; return ADK_OK; }
                if (rc != ADK_ERR_STATE) { g_err = "op " + std::to_string(i) + " (fused): " + g_err; return rc; }
            }
        }
        rc = op_conv_args(p, i, frames, ext, a);
        if (rc == ADK_OK) rc = run_conv(a, o.impl, s, p->ws);
    } else if (o.kind == ADK_OP_MEAN) {
        RingMeanArgs m;
        memset(&m, 0, sizeof(m));
        adk_ring_view out = view_of(p, o.out_ring, frames, ext, 0);
        m.out = out.base; m.out_rows = out.rows; m.out_cursor = out.cursor;
        m.n = o.n_mean; m.channels = out.channels; m.batch = p->batch; m.t = frames * p->rings[o.out_ring].rate;
        for (int k = 0; k < o.n_mean; ++k) {
            adk_ring_view sv = view_of(p, o.mean_rings[k], frames, ext, 0);
            m.src[k] = sv.base; m.src_rows[k] = sv.rows; m.src_cursor[k] = sv.cursor;
        }
        rc = launch_ring_mean(m, s);
    } else if (o.kind == ADK_OP_HIST_REPLICATE) {
        if (p->fresh) {
            adk_ring_view v = view_of(p, o.in_ring, frames, ext, 0);
            rc = launch_hist_replicate(v.base, v.rows, v.channels, v.cursor, p->rings[o.in_ring].hist, p->batch, s);
        }
    } else {
        ConvArgs a;
        if (consumed && max_consume >= 2 && ext[o.ext_src] && op_write_conv_fusable(p, i, frames, ext, a)) {
            rc = launch_conv_cin1_write(a, static_cast<const float*>(ext[o.ext_src]), s);
            if (rc == ADK_OK) { *consumed = 2; return ADK_OK; }
            if (rc != ADK_ERR_STATE) { g_err = "op " + std::to_string(i) + " (ring write + conv): " + g_err; return rc; }
        }
        adk_ring_view out = view_of(p, o.out_ring, frames, ext, 0);
        const float* mean = o.mean_off >= 0 ? p->weights + o.mean_off : nullptr;
        const float* scale = o.scale_off >= 0 ? p->weights + o.scale_off : nullptr;
        rc = adk_ring_write(static_cast<const float*>(ext[o.ext_src]), out, mean, scale, p->batch,
                            frames * p->rings[o.out_ring].rate, s);
    }
    if (rc != ADK_OK) g_err = "op " + std::to_string(i) + ": " + g_err;
    return rc;
}

static bool op_touches_ext(const adk_program* p, const adk_op_desc& o) {
    auto ext_ring = [&](int id) { return id >= 0 && p->rings[id].external >= 0; };
    if (o.kind == ADK_OP_RING_WRITE) return true;
    if (o.kind == ADK_OP_MEAN) {
        for (int k = 0; k < o.n_mean; ++k) if (ext_ring(o.mean_rings[k])) return true;
        return ext_ring(o.out_ring);
    }
    return ext_ring(o.in_ring) || ext_ring(o.out_ring) || ext_ring(o.res_ring);
}

// Phase of the cursor state among the states full-size steps cycle through, or -1 (a short step, a restored snapshot from
// another schedule ...): ring i of a program that has only ever taken max_frames-sized steps sits at (k * adv_i) mod rows_i.
static int graph_phase(const adk_program* p) {
    for (int k = 0; k < p->period; ++k) {
        bool ok = true;
        for (size_t i = 0; i < p->rings.size() && ok; ++i) {
            if (p->rings[i].external >= 0) continue;
            const long long adv = (long long)p->max_frames * p->rings[i].rate;
            ok = p->cursor[i] == (int32_t)((k * adv) % p->rows[i]);
        }
        if (ok) return k;
    }
    return -1;
}

static void drop_graphs(adk_program* p) {
    for (hipGraphExec_t g : p->gexec) if (g) (void)hipGraphExecDestroy(g);
    p->gexec.assign(p->period > 0 ? p->period : 0, nullptr);
    p->seen.assign(p->period > 0 ? p->period : 0, 0);
}

extern "C" int adk_program_set_graph(adk_program* p, int32_t enabled) {
    if (!p) return fail(ADK_ERR_ARG, "program_set_graph: null program");
    if (!enabled) { p->graph = false; return ADK_OK; }
    // period of the cursor states under full-size steps = lcm_i rows_i / gcd(rows_i, adv_i); it is small only when the caller
    // sized the rings for it (history rounded up so that rows_i is a small multiple of adv_i: audiodec_amd/program.py)
    long long period = 1;
    auto gcd = [](long long a, long long b) { while (b) { long long t = a % b; a = b; b = t; } return a; };
    for (size_t i = 0; i < p->rings.size(); ++i) {
        if (p->rings[i].external >= 0) continue;
        const long long adv = (long long)p->max_frames * p->rings[i].rate, rows = p->rows[i];
        const long long pi = rows / gcd(rows, adv % rows == 0 ? rows : adv % rows);
        period = period / gcd(period, pi) * pi;
        if (period > 64) return fail(ADK_ERR_STATE, "program_set_graph: the ring cursors have no short period (size the rings as multiples of the step)");
    }
    // the ops touching caller buffers (first ring_write, last conv) stay eager: their pointers change from call to call
    int lo = 0, hi = (int)p->ops.size();
    while (lo < hi && op_touches_ext(p, p->ops[lo])) ++lo;
    while (hi > lo && op_touches_ext(p, p->ops[hi - 1])) --hi;
    for (int i = lo; i < hi; ++i)
        if (op_touches_ext(p, p->ops[i]) || p->ops[i].kind == ADK_OP_HIST_REPLICATE)
            return fail(ADK_ERR_STATE, "program_set_graph: an op in the middle of the sequence uses a caller buffer (or the offline history replicate)");
    if (hi - lo < 2) return fail(ADK_ERR_STATE, "program_set_graph: nothing to capture");
    DeviceGuard guard(p->device);
    p->period = (int)period; p->g_lo = lo; p->g_hi = hi;
    drop_graphs(p);
    p->graph = true;
    return ADK_OK;
}

extern "C" int adk_program_graph_stats(const adk_program* p, int64_t* replays, int64_t* captures, int32_t* period) {
    if (!p) return fail(ADK_ERR_ARG, "program_graph_stats: null program");
    if (replays) *replays = p->replays;
    if (captures) *captures = p->captures;
    if (period) *period = p->graph ? p->period : 0;
    return ADK_OK;
}

static int program_step(adk_program* p, int32_t frames, void* const* ext, int32_t n_ext, void* stream);

extern "C" int adk_program_step(adk_program* p, int32_t frames, void* const* ext, int32_t n_ext, void* stream) {
    if (!p) return fail(ADK_ERR_ARG, "program_step: null program");
    p->replaying = false;
    return program_step(p, frames, ext, n_ext, stream);
}

extern "C" int adk_program_step_ex(adk_program* p, int32_t frames, void* const* ext, int32_t n_ext, void* stream, int32_t step_flags) {
    if (!p) return fail(ADK_ERR_ARG, "program_step_ex: null program");
    if (step_flags & ~ADK_STEP_REPLAY) return fail(ADK_ERR_ARG, "program_step_ex: unknown step flag");
    p->replaying = (step_flags & ADK_STEP_REPLAY) != 0;
    const int rc = program_step(p, frames, ext, n_ext, stream);
    p->replaying = false;
    return rc;
}

static int program_step(adk_program* p, int32_t frames, void* const* ext, int32_t n_ext, void* stream) {
    if (frames <= 0 || frames > p->max_frames) return fail(ADK_ERR_SHAPE, "program_step: frames must be in [1, max_frames]");
    if (n_ext < p->n_ext || (p->n_ext > 0 && !ext)) return fail(ADK_ERR_ARG, "program_step: missing external buffers");
    for (int i = 0; i < p->n_ext; ++i) {
        if (ext[i]) continue;
        bool needed = !p->replaying;                       // on a replay the sources of the ring writes may be absent
        for (size_t k = 0; k < p->ops.size() && !needed; ++k) {
            const adk_op_desc& o = p->ops[k];
            auto is_ext = [&](int id) { return id >= 0 && p->rings[id].external == i; };
            if (o.kind == ADK_OP_RING_WRITE) continue;
            if (o.kind == ADK_OP_MEAN) { for (int q = 0; q < o.n_mean; ++q) needed |= is_ext(o.mean_rings[q]); needed |= is_ext(o.out_ring); }
            else needed |= is_ext(o.in_ring) || is_ext(o.out_ring) || is_ext(o.res_ring);
        }
        if (needed) return fail(ADK_ERR_ARG, "program_step: null external buffer");
    }
    hipStream_t s = static_cast<hipStream_t>(stream);
    DeviceGuard guard(p->device);            // launches, the workspace and the flag word belong to the program's device
    const int n_ops = (int)p->ops.size();
    if (p->profiling && (int)p->ev.size() != n_ops + 1) {
        for (hipEvent_t e : p->ev) (void)hipEventDestroy(e);
        p->ev.resize(n_ops + 1);
        for (auto& e : p->ev) ADK_HIP_CHECK(hipEventCreate(&e));
    }
    // HIP-graph replay: full-size steps of a warmed-up program in a known cursor phase (see adk_program_set_graph)
    const int phase = (p->graph && !p->profiling && !p->fresh && frames == p->max_frames) ? graph_phase(p) : -1;
    const bool replay = phase >= 0 && p->seen[phase] && s != nullptr;      // the legacy default stream cannot be captured: eager there
    if (p->profiling) ADK_HIP_CHECK(hipEventRecord(p->ev[0], s));
    for (int i = 0; i < n_ops; ++i) {
        if (replay && i == p->g_lo) {
            if (!p->gexec[phase]) {
                // capture ops [g_lo, g_hi) once for this phase.  The stream-K publish flags carry per-launch epochs, which are
                // frozen in a captured launch: the flags are zeroed by a memset node at the head of the graph instead (a stale
                // flag of the previous replay would otherwise pass for this replay's publish).
                hipGraph_t g = nullptr;
                ADK_HIP_CHECK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
                int rc = ADK_OK;
                hipError_t e = hipMemsetAsync(reinterpret_cast<char*>(p->ws.ptr) + p->ws.flags_offset, 0, p->ws.bytes - p->ws.flags_offset, s);
                if (e != hipSuccess) rc = fail(ADK_ERR_HIP, std::string("graph capture: memset: ") + hipGetErrorString(e));
                // (no fusion across the end of the captured range: op g_hi touches a caller buffer whose pointer would be baked into the graph)
                for (int k = p->g_lo; k < p->g_hi && rc == ADK_OK;) { int used = 1; rc = run_op(p, k, frames, ext, s, p->g_hi - k, &used); k += used; }
                e = hipStreamEndCapture(s, &g);
                if (rc != ADK_OK) { if (g) (void)hipGraphDestroy(g); return rc; }
                if (e != hipSuccess || !g) return fail(ADK_ERR_HIP, std::string("hipStreamEndCapture: ") + hipGetErrorString(e));
                hipGraphExec_t ge = nullptr;
                e = hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
                (void)hipGraphDestroy(g);
                if (e != hipSuccess) return fail(ADK_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
                p->gexec[phase] = ge;
                ++p->captures;
            }
            ADK_HIP_CHECK(hipGraphLaunch(p->gexec[phase], s));
            ++p->replays;
            i = p->g_hi - 1;
            continue;
        }
        int used = 1;
        // a launch may cover the ops that follow (residual unit / chain), but never reach into a range that is replayed as a graph
        int rc = run_op(p, i, frames, ext, s, (replay && i < p->g_lo ? p->g_lo : n_ops) - i, &used);
        if (rc != ADK_OK) return rc;
        if (p->profiling) ADK_HIP_CHECK(hipEventRecord(p->ev[i + 1], s));
        for (; used > 1; --used) {            // the successors ran inside the same launch
            ++i;
            if (p->profiling) ADK_HIP_CHECK(hipEventRecord(p->ev[i + 1], s));
        }
    }
    if (phase >= 0) p->seen[phase] = 1;
    for (size_t i = 0; i < p->rings.size(); ++i)
        if (p->rings[i].external < 0)
            p->cursor[i] = (int32_t)(((long long)p->cursor[i] + (long long)frames * p->rings[i].rate) % p->rows[i]);
    p->fresh = false;
    ++p->steps_since_reset;
    return ADK_OK;
}

extern "C" int adk_program_describe_op(adk_program* p, int32_t op, int32_t frames, char* buf, int32_t n) {
    if (!p || !buf || n <= 0 || op < 0 || op >= (int)p->ops.size()) return fail(ADK_ERR_ARG, "program_describe_op: bad arguments");
    const adk_op_desc& o = p->ops[op];
    std::string name = o.kind == ADK_OP_MEAN ? "ring_mean" : (o.kind == ADK_OP_HIST_REPLICATE ? "hist_replicate" : "ring_write");
    alignas(16) static float aligned_dummy0[4];
    void* ext0[8];
    for (auto& e : ext0) e = aligned_dummy0;
    ConvArgs wa;
    if (o.kind == ADK_OP_RING_WRITE && op_write_conv_fusable(p, op, frames, ext0, wa)) name = "ring_write+conv_cin1";
    else if (o.kind == ADK_OP_CONV && op > 0 && op_write_conv_fusable(p, op - 1, frames, ext0, wa)) name = "(in the launch of the ring write)";
    else if (o.kind == ADK_OP_CONV) {
        adk_conv_desc d = o.conv;
        d.w = o.w_off >= 0 ? p->weights + o.w_off : nullptr;
        d.w_frag = o.wf_off >= 0 ? p->weights + o.wf_off : nullptr;
        d.bias = o.b_off >= 0 ? p->weights + o.b_off : nullptr;
        alignas(16) static float aligned_dummy[4];   // stand-in for the (16-byte aligned) external buffers:
        void* ext[8];                                // only geometry/alignment is inspected, nothing is launched
        for (auto& e : ext) e = aligned_dummy;
        adk_ring_view in = view_of(p, o.in_ring, frames, ext, o.in_ch_off);
        adk_ring_view out = view_of(p, o.out_ring, frames, ext, o.out_ch_off);
        adk_ring_view res; memset(&res, 0, sizeof(res));
        if (o.res_ring >= 0) res = view_of(p, o.res_ring, frames, ext, o.res_ch_off);
        ConvArgs a;
        int rc = build_args(d, in, out, res, p->batch, frames * o.rate_out, a);
        if (rc != ADK_OK) return rc;
        if (o.in_shadow > 0) a.in_sh = view_of(p, o.in_shadow - 1, frames, ext, 0).base;      // (kernel choice looks at the shadows)
        if (o.out_shadow > 0) a.out_sh = view_of(p, o.out_shadow - 1, frames, ext, 0).base;
        name = conv_kernel_name(a, o.impl);
        ConvArgs f1, f2;
        ConvArgs c[kMaxChain]; int keep[kMaxChain];
        int head = -1;                              // the chain this op belongs to, when that chain runs as one launch
        for (int h = op; h >= 0 && h > op - kMaxChain; --h)
            if (p->ops[h].kind == ADK_OP_CONV && p->ops[h].chain >= 2 && h + p->ops[h].chain > op) { head = h; break; }
        if (head >= 0 && op_chain_fusable(p, head, frames, ext, c, keep)) name = head == op ? conv_rb16_name(c, p->ops[head].chain) : "(fused into the previous op)";
        else if (o.fuse_next && op_pair_kind(p, op, frames, ext, f1, f2)) { const int k = op_pair_kind(p, op, frames, ext, f1, f2); name = k == 2 ? "conv_ou16<192>" : (k == 3 ? "conv_oc16<96>" : (f1.cin_g == 32 ? "conv_rl16_unit<32>" : "conv_rl16_unit<64>")); }
        else if (op > 0 && p->ops[op - 1].fuse_next && op_pair_kind(p, op - 1, frames, ext, f1, f2)) name = "(fused into the previous op)";
    }
    snprintf(buf, n, "%s", name.c_str());
    return ADK_OK;
}

extern "C" int adk_program_reset(adk_program* p, void* stream) {
    if (!p) return fail(ADK_ERR_ARG, "program_reset: null program");
    DeviceGuard guard(p->device);
    for (size_t i = 0; i < p->rings.size(); ++i) {
        const adk_ring_desc& r = p->rings[i];
        if (r.external >= 0) continue;
        ADK_HIP_CHECK(hipMemsetAsync(p->arena + r.arena_off, 0, sizeof(float) * (size_t)p->batch * p->rows[i] * r.channels,
                                     static_cast<hipStream_t>(stream)));
        p->cursor[i] = 0;
    }
    p->fresh = true;
    p->steps_since_reset = 0;
    return ADK_OK;
}

extern "C" int adk_program_flags(adk_program* p, void* stream, int32_t* out) {
    if (!p || !out) return fail(ADK_ERR_ARG, "program_flags: null argument");
    DeviceGuard guard(p->device);
    int v = 0;
    const int rc = program_fetch_clear(p, static_cast<hipStream_t>(stream), &v);
    *out = v;
    return rc;
}

extern "C" int adk_program_flags_post(adk_program* p, void* stream, int64_t* ticket) {
    if (!p || !ticket) return fail(ADK_ERR_ARG, "program_flags_post: null argument");
    DeviceGuard guard(p->device);
    const int slot = (int)(p->post_next % ADK_POST_SLOTS);
    if (!p->post_ev[slot]) ADK_HIP_CHECK(hipEventCreateWithFlags(&p->post_ev[slot], hipEventDisableTiming));
    // the post IS the event: the program's flag word is pinned host memory the kernels report into directly (rvq.hip: flag pool) -- once the
    // event has completed, everything the step's launches had to say is in it.  (Rounds 4-5: a 1-thread kernel that moved a device word.)
    ADK_HIP_CHECK(hipEventRecord(p->post_ev[slot], static_cast<hipStream_t>(stream)));
    *ticket = p->post_next++;
    return ADK_OK;
}

extern "C" int adk_program_flags_poll(adk_program* p, int64_t ticket, int32_t block, int32_t* done, int32_t* flags) {
    if (!p || !done || !flags) return fail(ADK_ERR_ARG, "program_flags_poll: null argument");
    if (ticket < 0 || ticket >= p->post_next || ticket + ADK_POST_SLOTS < p->post_next)
        return fail(ADK_ERR_STATE, "program_flags_poll: unknown ticket, or more than ADK_POST_SLOTS posts since it was issued");
    const int slot = (int)(ticket % ADK_POST_SLOTS);
    DeviceGuard guard(p->device);
    *done = 0; *flags = 0;
    if (block) {
        ADK_HIP_CHECK(hipEventSynchronize(p->post_ev[slot]));
    } else {
        const hipError_t e = hipEventQuery(p->post_ev[slot]);
        if (e == hipErrorNotReady) { (void)hipGetLastError(); return ADK_OK; }
        if (e != hipSuccess) return fail(ADK_ERR_HIP, std::string("hipEventQuery: ") + hipGetErrorString(e));
    }
    *done = 1;
    int v = 0;
    const int rc = flag_pool_take(p->device, p->flags, &v);       // read AND clear: what the program has reported up to now
    *flags = v;
    return rc;
}

extern "C" int adk_program_rewind(adk_program* p, int32_t frames) {
    if (!p) return fail(ADK_ERR_ARG, "program_rewind: null program");
    if (frames <= 0 || frames > p->max_frames) return fail(ADK_ERR_SHAPE, "program_rewind: frames must be in [1, max_frames]");
    for (size_t i = 0; i < p->rings.size(); ++i)
        if (p->rings[i].external < 0) {
            const long long back = ((long long)frames * p->rings[i].rate) % p->rows[i];
            p->cursor[i] = (int32_t)(((long long)p->cursor[i] - back + p->rows[i]) % p->rows[i]);
        }
    // the repeated step is the first one after a reset iff the rewound one was -- also after several rewinds in a row (ABI 13: up to
    // extra_rows / step of them): a counter, not one saved level
    if (p->steps_since_reset > 0) --p->steps_since_reset;
    p->fresh = p->steps_since_reset == 0;
    return ADK_OK;
}

extern "C" int adk_program_get_fresh(const adk_program* p) { return p ? (p->fresh ? 1 : 0) : fail(ADK_ERR_ARG, "program_get_fresh: null program"); }

extern "C" int adk_program_set_fresh(adk_program* p, int32_t fresh) {
    if (!p) return fail(ADK_ERR_ARG, "program_set_fresh: null program");
    p->fresh = fresh != 0;
    p->steps_since_reset = p->fresh ? 0 : (p->steps_since_reset > 0 ? p->steps_since_reset : (1ll << 40));   // "not fresh": no number of rewinds makes it fresh
    return ADK_OK;
}

extern "C" int adk_program_get_cursors(const adk_program* p, int32_t* cursors, int32_t n) {
    if (!p || !cursors || n != (int)p->cursor.size()) return fail(ADK_ERR_ARG, "program_get_cursors: bad arguments");
    memcpy(cursors, p->cursor.data(), sizeof(int32_t) * n);
    return ADK_OK;
}

extern "C" int adk_program_set_cursors(adk_program* p, const int32_t* cursors, int32_t n) {
    if (!p || !cursors || n != (int)p->cursor.size()) return fail(ADK_ERR_ARG, "program_set_cursors: bad arguments");
    for (int i = 0; i < n; ++i)
        if (p->rings[i].external < 0 && (cursors[i] < 0 || cursors[i] >= p->rows[i]))
            return fail(ADK_ERR_SHAPE, "program_set_cursors: cursor outside ring");
    memcpy(p->cursor.data(), cursors, sizeof(int32_t) * n);
    return ADK_OK;
}

extern "C" int adk_program_set_workgroups(adk_program* p, int32_t workgroups) {
    if (!p) return fail(ADK_ERR_ARG, "program_set_workgroups: null program");
    if (workgroups < 0 || (workgroups > 0 && workgroups < 8)) return fail(ADK_ERR_SHAPE, "program_set_workgroups: 0 (whole chip) or >= 8");
    if (p->ws.workgroups != workgroups / 8 * 8) {
        DeviceGuard guard(p->device);
        drop_graphs(p);                       // the captured launches baked the old share into their grids
    }
    p->ws.workgroups = workgroups / 8 * 8;
    return ADK_OK;
}

extern "C" int adk_program_set_profiling(adk_program* p, int32_t enabled) {
    if (!p) return fail(ADK_ERR_ARG, "program_set_profiling: null program");
    p->profiling = enabled != 0;
    return ADK_OK;
}

extern "C" int adk_program_last_op_ms(adk_program* p, float* ms, int32_t n) {
    if (!p || !ms || n != (int)p->ops.size()) return fail(ADK_ERR_ARG, "program_last_op_ms: bad arguments");
    if (!p->profiling || (int)p->ev.size() != n + 1) return fail(ADK_ERR_STATE, "program_last_op_ms: profiling not enabled / no step yet");
    DeviceGuard guard(p->device);
    ADK_HIP_CHECK(hipEventSynchronize(p->ev[n]));
    for (int i = 0; i < n; ++i) ADK_HIP_CHECK(hipEventElapsedTime(&ms[i], p->ev[i], p->ev[i + 1]));
    return ADK_OK;
}
