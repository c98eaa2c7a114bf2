/*
 * audiodec_hip.h -- C ABI of libaudiodec_hip.so: the AudioDec streaming hot path on MI355X (gfx950).
 *
 * The reference (facebookresearch/AudioDec) has no native layer: its hot path is PyTorch module
 * methods.  This header is the FFI a maintainer would bind in their place (ctypes stub in
 * INTEGRATION.md); every entry point names the reference method(s) it replaces.
 *
 * Conventions
 *   - the matrix-core conv kernel keeps split-tile partial sums in a scratch workspace: a program owns
 *     one (allocated in adk_program_create); op-level adk_causal_conv uses a per-host-thread one that
 *     is allocated on that thread's first call
 *   - all data pointers are DEVICE pointers (fp32 unless stated); `stream` is a hipStream_t passed
 *     as void*; calls enqueue work on that stream and return without synchronising
 *   - return value: 0 = ADK_OK, negative = error (text via adk_last_error(), thread-local)
 *   - no hidden device allocation in step/op calls; a handle may be used by one host thread at a
 *     time, different handles concurrently (reference threading model: bin/stream.py:212-239)
 *   - devices: every call runs on the HIP device that OWNS its buffers, whatever device the calling thread has
 *     current (the reference's --tx_cuda / --rx_cuda put the two halves on different GPUs, demoStream.py:33-40):
 *     a program is bound to the device of its arena at adk_program_create; op-level calls look the device up from
 *     their output pointer; the scratch workspace and the sticky flag word are per device
 *
 * Data layout ("rings")
 *   Every stateful conv input lives in a ring of channel-last rows: ring[b][r][c], r in [0,rows),
 *   one row = one time step of `channels` floats.  The reference's per-layer `pad_buffer`
 *   (layers/conv_layer.py:144-146,153-156) is the `hist` rows in front of the cursor; producers write
 *   new rows at the cursor, consumers read [cursor-hist, cursor+T); nothing is ever copied or
 *   shifted.  The ring stores the RAW (pre-activation) signal; the consumer's input activation
 *   (ELU / LeakyReLU) is applied while the tile is staged on chip.
 */
#ifndef AUDIODEC_HIP_H
#define AUDIODEC_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ADK_ABI_VERSION 14

enum { ADK_OK = 0, ADK_ERR_ARG = -1, ADK_ERR_SHAPE = -2, ADK_ERR_HIP = -3, ADK_ERR_STATE = -4 };

/* activations: layers/activation_function.py:18-22 -> torch.nn.{ELU,LeakyReLU,Tanh} */
enum { ADK_ACT_NONE = 0, ADK_ACT_ELU = 1, ADK_ACT_LEAKY = 2, ADK_ACT_TANH = 3 };

/* kernel selection for adk_causal_conv (ADK_IMPL_AUTO in production; others for tests/benchmarks) */
enum { ADK_IMPL_AUTO = 0, ADK_IMPL_DIRECT = 1, ADK_IMPL_MFMA = 2 /* stream-K implicit GEMM */, ADK_IMPL_MFMA_ROWS = 3 /* rows-in-LDS */,
       /* Opt-in split-precision kernels: every f32 operand is carried as f16 hi + f16 lo/2048 and a product sum is
          formed from three f16 MFMAs (hi*hi, hi*lo, lo*hi) with f32 accumulation -- measured error vs fp64 is
          below that of the f32 MFMA chain (profiles/r1_f16_split_probe.txt).  w_frag must then hold the
          adk_pack_weights_split16 layout.  |operand| > 65504 (-> non-finite outputs) raises device flag bit 3.
          SPLIT16 picks between the rows-in-LDS and the stream-K variant like AUTO does; the other two force one. */
       ADK_IMPL_SPLIT16 = 4, ADK_IMPL_SPLIT16_ROWS = 5, ADK_IMPL_SPLIT16_SK = 6,
       /* the streaming kernel of the last up-sampling stage (fused activation -> ConvTranspose1d 64 -> Cout, s*Cout <= 96,
          + bias; HiFiGAN.py:285-289): SPLIT16 picks it for that layer shape, this value forces it */
       ADK_IMPL_SPLIT16_UP = 7 };

const char* adk_last_error(void);
int adk_abi_version(void);
/* sticky device-side flags since the last call (bit 0: adk_rvq_lookup saw an out-of-range index,
 * where F.embedding would raise; bit 1: a stream-K conv workgroup gave up waiting for another workgroup's
 * partial tile -- results of that launch are invalid; bit 2: adk_codes_pack saw an index that
 * is not a code of its stage; bit 3: a split-f16 kernel produced a non-finite output -- an operand beyond the f16 range (|v| > 65504) or a
 * non-finite input; results of that launch are invalid); the value is the OR over every device the library
 * has launched on and every program on them (one sweep kernel per device over the pool the program words live in);
 * reading synchronises those devices and clears the words.  The Python facade turns a set bit
 * into an exception at its synchronisation points (audiodec_amd/native.py: raise_on_device_flags) */
int adk_debug_flags(int32_t* out);
/* tuning hook: force the MFMA conv tile config (0..6), -1 = heuristic (also env ADK_CONV_CFG) */
int adk_set_conv_cfg(int32_t cfg);
/* tuning / test hook: named process-wide integer options, read at every launch decision.
 *   "chain_max_channels"  residual chains (adk_op_desc.chain) run as one launch only up to this many channels per group
 *                         (default 128 = every chain the kernel takes; 64: not the 128-channel ones; 0: never -- the ops of a
 *                         chain are then launched one by one, as with ADK_CHAIN=0).  State rings are compatible either way:
 *                         the value may change between two steps of a running program.
 *   "chain_min_channels"  ... and from this many channels per group (default 0)
 *   "chain_min_blocks"    ... and only for launches of at least this many (stream, group) pairs (default 0: always.  Until round 4 the
 *                         default was 160 -- with the chain kernels of that time the per-op launches, which spread a stream's time tiles
 *                         over many CUs, were faster below it; now the chain is faster at every launch size: 1 stream 0.74 -> 0.71 ms per
 *                         frame, 48 streams 0.86 -> 0.77 ms, profiles/r4_few_streams.md)
 *   "rvq_rows"            rows per workgroup of the residual-VQ search when dim == 64, size == 1024 (csrc/rvq.hip, rvq_encode_v4): from
 *                         "rvq_v4_min" rows (default 192) on, 2 or 4 rows share a workgroup's code registers; 1 (default) = 2 up to 512
 *                         rows, 4 above; 0: the round-3 kernels at every row count.  Indices and zq are bit-identical either way.
 *   "rvq_v4_min"          see above
 *   "gv16_max_columns"    convs of at most this many columns (streams x steps per call) run as conv_gv16 (csrc/conv_mfma.hip: one wave per
 *                         32-row x 32-column output tile and K slice, no LDS) instead of the stream-K kernel; default 32, 0 = never
 *                         (also env ADK_GV16_MAXN).  Results of the two kernels agree to f32 round-off, not bit for bit.
 *   "conv_ou16", "conv_oc16", "conv_cin1w"   0: the op pairs these launches cover run as two launches again (defaults 1; also env ADK_CONV_OU16,
 *                         ADK_CONV_OC16, ADK_CONV_CIN1W): conv_out + the last up-sampler (csrc/conv_ou16.hip: f32 round-off against the two-launch
 *                         form), the last conv_out + the output conv (csrc/conv_oc16.hip), the encoder's ring write + its Cin = 1 conv
 *                         (csrc/conv_direct.hip: bit-identical).  A/B measurements and the tests that compare the two forms.
 * ADK_ERR_ARG for an unknown name. */
int adk_set_option(const char* name, int32_t value);
/* Introspection of the stream-K launch schedule (pure host logic, no device needed; used by the CPU tests): a matrix-core conv
 * launch over `tiles` output tiles of `chunks` 64-deep K chunks each, allowed at most `cap` persistent workgroups (0: the library
 * default), runs plan[0] workgroups ("ranges").  plan[1] > 0: every tile is cut into plan[1] ranges (plan[1] == 2: the first takes
 * plan[3] chunks); plan[2] > 0: every range takes plan[2] whole tiles; both 0: tiles * chunks / plan[0] work units each, wherever
 * that cuts.  adk_streamk_range_start gives the first work unit (tile * chunks + chunk) of range r, r = plan[0]: one past the last;
 * -1 on a bad argument. */
int adk_streamk_plan(int64_t tiles, int32_t chunks, int32_t cap, int32_t* plan /* [4] */);
int64_t adk_streamk_range_start(int64_t tiles, int32_t chunks, const int32_t* plan, int32_t r);

/* A view of one ring for one call. */
typedef struct {
    float*  base;      /* stream b starts at base + b * rows * channels                            */
    int32_t rows;      /* ring length R                                                             */
    int32_t channels;  /* row length C (all groups)                                                 */
    int32_t cursor;    /* row of the first NEW time step of this call, 0 <= cursor < rows           */
    int32_t ch_off;    /* first channel this op touches                                             */
} adk_ring_view;

/*
 * One fused causal convolution:  replaces
 *   CausalConv1d.inference            layers/conv_layer.py:153-156  (up == 1)
 *   CausalConvTranspose1d.inference   layers/conv_layer.py:194-197  (up == stride s; polyphase form:
 *       taps = 2, cout_g = s*Cout, weight row (r*Cout+co) = [W[:,co,s+r] | W[:,co,r]])
 *   Conv1d1x1                         layers/conv_layer.py:28-32    (taps == 1, hist == 0)
 * plus the element-wise ops the reference runs around them: input activation, bias, residual add
 * (residual_unit.py:78-81, residual_block.py:99-105), output activation (HiFiGAN.py:294-296).
 *
 * For stream b, output step t in [0, t_out), GEMM row m in [0, groups*cout_g), g = m / cout_g:
 *   acc = bias[m] + sum_{j<taps} sum_{ci<cin_g}
 *           w[m][j*cin_g + ci] * act_in( in[b][(in.cursor - hist + t*stride + j*dilation) mod R][in.ch_off + g*in_group_stride + ci] )
 *   acc += res[b][(res.cursor + t) mod R][res.ch_off + g*res_group_stride + (m mod cout_g)]     (if res.base)
 *   out[b][(out.cursor + t*up + m / cout_real) mod R][out.ch_off + m mod cout_real] = act_out(acc)
 */
typedef struct {
    int32_t cin_g, cout_g, groups;       /* per-group channels                                       */
    int32_t taps, stride, dilation;      /* stride = input rows per output step                      */
    int32_t hist;                        /* history rows in front of in.cursor: (taps-1)*dilation    */
    int32_t up, cout_real;               /* up == 1: cout_real = groups*cout_g; transposed: see above */
    int32_t in_group_stride;             /* 0 when all groups read the same channels (x.repeat, multi_fusion.py:134) */
    int32_t res_group_stride;
    int32_t act_in;  float act_in_slope;
    int32_t act_out;
    const float* w;                      /* [groups*cout_g][taps*cin_g], tap-major / channel-minor; may be NULL when
                                            w_frag is given and the MFMA kernel is taken             */
    const float* w_frag;                 /* the same weights in MFMA-fragment order (adk_pack_weights_mfma) or NULL:
                                            without it only the VALU kernel can run                  */
    const float* bias;                   /* [groups*cout_g] or NULL                                  */
} adk_conv_desc;

/* Re-pack row-major conv weights [groups*cout_g][ktot] for the matrix-core kernel:
 * out[g][m-tile of 32][k-group of 8][lane 0..63][4] with lane (i = lane&31, h = lane>>5) holding
 * W[32*mt + i][8*kg + 4*h + 0..3]; rows beyond cout_g and the K tail (K is padded to a multiple
 * of 64) are zero.  out needs
 * adk_packed_weight_floats(groups, cout_g, ktot) floats; ktot % 8 == 0.  Done once at load time. */
int64_t adk_packed_weight_floats(int32_t groups, int32_t cout_g, int32_t ktot);
int adk_pack_weights_mfma(const float* w, float* out, int32_t groups, int32_t cout_g, int32_t ktot, void* stream);
/* Split-f16 fragment order for ADK_IMPL_*_SPLIT16: out[g][m-tile of 32][16-k chunk][hi | lo][lane 0..63][8 x f16]
 * with lane (i = lane&31, h = lane>>5) holding W[32*mt + i][16*chunk + 8*h + 0..7], hi = f16(W),
 * lo = f16((W - hi) * 2048); rows beyond cout_g are zero.  ktot % 16 == 0; out needs
 * adk_packed_weight_floats_split16(groups, cout_g, ktot) floats (= 32*mt32 * ktot per group). */
int64_t adk_packed_weight_floats_split16(int32_t groups, int32_t cout_g, int32_t ktot);
int adk_pack_weights_split16(const float* w, float* out, int32_t groups, int32_t cout_g, int32_t ktot, void* stream);

int adk_causal_conv(const adk_conv_desc* d, adk_ring_view in, adk_ring_view out, adk_ring_view res,
                    int32_t batch, int32_t t_out, int32_t impl, void* stream);
/* name of the kernel the call above would launch (e.g. "conv_sk16<128x64>"), nothing is launched: for profiles and
 * for tests that must know which kernel they exercised */
int adk_causal_conv_describe(const adk_conv_desc* d, adk_ring_view in, adk_ring_view out, adk_ring_view res,
                             int32_t batch, int32_t t_out, int32_t impl, char* buf, int32_t n);
/* timing aid (bench.py, tools): `iters` back-to-back launches of the call above on `stream`, bracketed by HIP events recorded
 * on that stream; synchronises; *avg_us = average microseconds per launch (no host language in the loop) */
int adk_causal_conv_time(const adk_conv_desc* d, adk_ring_view in, adk_ring_view out, adk_ring_view res,
                         int32_t batch, int32_t t_out, int32_t impl, int32_t iters, void* stream, float* avg_us);

/*
 * Copy caller rows into a ring, optionally normalising: ring row = (src - mean) / scale.
 * Replaces the torch.cat of new samples onto the state (conv_layer.py:154) for the first layer and
 * HiFiGAN StreamGenerator.decode_norm (models/vocoder/HiFiGAN.py:276-279; true division).
 * src is [batch][t][channels] contiguous.
 */
int adk_ring_write(const float* src, adk_ring_view ring, const float* mean, const float* scale,
                   int32_t batch, int32_t t, void* stream);

/*
 * Residual VQ encode: replaces ResidualVQ.forward_index(flatten_idx=True) over
 * VectorQuantize.forward_index (layers/vq_module.py:136-149, 90-104) for n_rows = B*T rows.
 *   z      [n_rows][dim]            (channel-last latent, i.e. Quantizer.encode's z.transpose(2,1))
 *   embed  [n_q][dim][size]         the reference's `embed` buffers (codes are columns)
 *   enorm  [n_q][size]              embed.pow(2).sum(0) (vq_module.py:96), computed once at load
 *   idx    [n_q][n_rows] int64      emitted index = code + size*stage (vq_module.py:145-146)
 *   zq     [n_rows][dim] or NULL    sum of the straight-through quantised vectors (quantized_out)
 * Arithmetic per stage follows the reference literally: dist = (|r|^2 - (2r).E) + |E|^2, argmax of
 * -dist with lowest index on ties, q' = r + (q - r), r <- r - q'.
 */
int adk_rvq_encode(const float* z, const float* embed, const float* enorm, int64_t* idx, float* zq,
                   int32_t n_rows, int32_t n_q, int32_t dim, int32_t size, void* stream);

/*
 * Residual VQ lookup: replaces ResidualVQ.lookup (layers/vq_module.py:159-161).
 *   idx [n_q][n_rows] int64 (global indices 0..n_q*size-1), codebook [n_q*size][dim]
 *   zq  [n_rows][dim] = sum over stages, stage 0 first
 * An out-of-range index reads code 0 and raises bit 0 of adk_debug_flags().
 */
int adk_rvq_lookup(const int64_t* idx, const float* codebook, float* zq,
                   int32_t n_rows, int32_t n_q, int32_t dim, int32_t n_codes, void* stream);

/*
 * Bit-packed code wire format (the reference passes the int64 index tensor through a
 * queue.Queue, bin/stream.py:224,230, and never serialises it).  One frame of one stream = n_q codes of
 * `bits` bits, LSB-first: code q (= emitted index - size*q) occupies bits [q*bits, (q+1)*bits) of the
 * adk_codes_frame_bytes(n_q, bits) = ceil(n_q*bits/8) byte frame; 8 x 10 bit = 10 bytes = 12.8 kbps at
 * 160 frames/s.  idx is [n_q][n_rows] int64 as emitted by adk_rvq_encode; payload is [n_rows][frame_bytes].
 * adk_codes_lookup = unpack fused into ResidualVQ.lookup (layers/vq_module.py:159-161).
 * A code >= size raises adk_debug_flags bit 2 (pack; it is packed as code 0 so that it cannot spill into the
 * neighbouring codes' bits) / bit 0 (lookup, reads code 0).
 */
int32_t adk_codes_frame_bytes(int32_t n_q, int32_t bits);
int adk_codes_pack(const int64_t* idx, uint8_t* payload, int32_t n_rows, int32_t n_q, int32_t bits, int32_t size, void* stream);
int adk_codes_unpack(const uint8_t* payload, int64_t* idx, int32_t n_rows, int32_t n_q, int32_t bits, int32_t size, void* stream);
int adk_codes_lookup(const uint8_t* payload, const float* codebook, float* zq, int32_t n_rows, int32_t n_q, int32_t bits,
                     int32_t size, int32_t dim, void* stream);

/* ------------------------------------------------------------------------------------------------
 * Pipeline level: a "program" is the fixed launch sequence of one model half (encoder+projector,
 * symmetric decoder, or HiFi-GAN vocoder) over B streams, with all rings in one caller-owned arena.
 * Replaces StreamGenerator.encode / decode (models/autoencoder/AudioDec.py:228-247,
 * models/vocoder/HiFiGAN.py:268-296) incl. the per-layer state hand-off.
 * ---------------------------------------------------------------------------------------------- */
typedef struct adk_program adk_program;

typedef struct {
    int32_t channels;    /* row length                                                               */
    int32_t hist;        /* max history any consumer needs                                           */
    int32_t rate;        /* rows per 'frame' (one hop of audio)                                      */
    int32_t external;    /* -1: lives in the arena; >= 0: index into step()'s ext[] (rows = frames*rate, cursor 0) */
    int64_t arena_off;   /* float offset of this ring in the arena (batch * rows * channels floats)  */
    int32_t extra_rows;  /* arena rings: rows beyond hist + max_frames * rate.  A ring with k * max_frames * rate extra rows still holds
                            the history of the step k steps back when the steps since have all been taken: adk_program_rewind can then be
                            applied k + 1 times in a row (the deferred guard of audiodec_amd/pipeline.py checks a step one to `depth` steps
                            late and repeats what followed it); 0 for external rings                                             */
    int32_t reserved_;   /* 0 */
} adk_ring_desc;

enum { ADK_OP_CONV = 0, ADK_OP_RING_WRITE = 1, ADK_OP_MEAN = 2,
       /* First step after create/reset only: copy the first new row of ring `in_ring` into its history rows.
          This is the replication left-pad of the NON-streaming CausalConvTranspose1d.forward
          (layers/conv_layer.py:189-192) that the offline drivers (codecTest.py / codecStatistic.py) run;
          streaming inference starts from a zero pad_buffer instead and never uses this op. */
       ADK_OP_HIST_REPLICATE = 3 };

typedef struct {
    int32_t kind;                        /* ADK_OP_*                                                  */
    int32_t in_ring, out_ring, res_ring; /* ring ids; res_ring = -1: none                             */
    int32_t in_ch_off, out_ch_off, res_ch_off;
    int32_t rate_out;                    /* output steps per frame (t_out = frames * rate_out)        */
    adk_conv_desc conv;                  /* conv.w / w_frag / bias ignored; use the offsets below     */
    int64_t w_off, wf_off, b_off;        /* float offsets into the weight blob (row-major weights, fragment-packed
                                            weights, bias); < 0: absent (at least one of w_off / wf_off)     */
    int64_t mean_off, scale_off;         /* ADK_OP_RING_WRITE: offsets of mean/scale, < 0: none       */
    int32_t ext_src;                     /* ADK_OP_RING_WRITE: index into ext[] of the source rows    */
    int32_t mean_rings[4];               /* ADK_OP_MEAN: out = (((r0 + r1) + r2) ...) / n over n_mean source rings
                                            (MultiReceptiveField: cs += block(c); c = cs / num_blocks,
                                            models/vocoder/modules/multi_fusion.py:73-79)              */
    int32_t n_mean;
    int32_t impl;                        /* ADK_IMPL_*                                                */
    int32_t fuse_next;                   /* 1: this conv's output ring is read only by the NEXT op, the 1x1 conv + residual of the same
                                            residual unit (residual_unit.py:78-81): the runner may launch both as one kernel
                                            (split-f16 rows-in-LDS kernel; the intermediate then never goes to memory) */
    int32_t chain;                       /* n >= 2: ops [this, this + n) are a residual chain -- units of (conv A; conv B + residual of A's
                                            input), each op reading the ring its predecessor writes, every intermediate ring read by nobody
                                            else (HiFiGANResidualBlock.inference, residual_block.py:99-105; the three CausalResidualUnits of
                                            an encoder / decoder block, encoder.py:76-81, decoder.py:73-78).  The runner may launch all n as
                                            ONE kernel (csrc/conv_rb16.hip: activations resident in LDS; an intermediate ring then receives
                                            only the rows later calls need as history).  0 / 1: no chain starts here */
    int32_t in_shadow, out_shadow;       /* 1 + id of the SHADOW ring of in_ring / out_ring, 0 = none (ADK_IMPL_SPLIT16* ops only).  A shadow
                                            ring has the geometry of its ring (channels, hist, rate, extra_rows; channels % 8 == 0) and holds, per
                                            8-channel group of a row (32 bytes, where the ring holds 8 floats), [8 x f16 hi][8 x f16 lo] of act(x):
                                            the split-f16 operand form the readers would otherwise recompute for every element they stage, once per
                                            tap and per 64-row tile of outputs.  out_shadow: this op writes it beside its output (stream-K kernel only:
                                            give such an op impl = ADK_IMPL_SPLIT16_SK; every op that writes the ring must carry it);
                                            in_shadow: the stream-K kernel stages from it instead of converting (other kernels ignore it:
                                            the ring itself is always complete).  Results are bit-identical with and without. */
    int32_t shadow_act;                  /* ADK_ACT_* the out_shadow carries = the act_in of the ring's readers */
    float shadow_slope;
} adk_op_desc;

/* rows of ring i = hist + max_frames * rate + extra_rows (arena rings). */
int adk_program_create(const adk_op_desc* ops, int32_t n_ops, const adk_ring_desc* rings, int32_t n_rings,
                       int32_t batch, int32_t max_frames, const float* weights, int64_t weights_floats,
                       float* arena, int64_t arena_floats, adk_program** out);
void adk_program_destroy(adk_program* p);
/* one call = `frames` hops for every stream; ext[i] are the external buffers named by the descs */
int adk_program_step(adk_program* p, int32_t frames, void* const* ext, int32_t n_ext, void* stream);
/* The same with options.  ADK_STEP_REPLAY: the ADK_OP_RING_WRITE ops are skipped -- the caller's input rows of this step are still in
 * their rings (a step that is being REPEATED after adk_program_rewind: the rows it wrote then are the rows it would write now), so the
 * repeat does not depend on the caller having kept its input buffer unchanged; ext[] entries read only by those ops may be NULL. */
enum { ADK_STEP_REPLAY = 1 };
int adk_program_step_ex(adk_program* p, int32_t frames, void* const* ext, int32_t n_ext, void* stream, int32_t step_flags);
/* reset_buffer(): zero all history (AudioDec.py:250-256, HiFiGAN.py:298-305) */
int adk_program_reset(adk_program* p, void* stream);
/* The launches of a program report device-side failures to a sticky word of the PROGRAM (same bits as adk_debug_flags, which
 * also collects the words of all live programs).  adk_program_flags waits for `stream`, returns the word and clears it (one
 * atomic exchange on the device): the caller that steps a program synchronously learns whether THIS step of THIS program
 * failed, whatever other programs run on other host threads / HIP streams.  Bit 3 (an operand of a split-f16 kernel left the
 * f16 range) is recoverable: a step only READS the history rows earlier steps left in the rings and WRITES this step's rows,
 * so adk_program_rewind(p, frames) -- cursors back by the `frames` hops of the step just taken -- followed by the same step
 * on a program lowered with the exact-f32 kernels over the same arena layout (copy the arena and the cursors across) repeats
 * it exactly; audiodec_amd/stream_generator.py does that automatically for synchronous callers ("guard"). */
int adk_program_flags(adk_program* p, void* stream, int32_t* out);
int adk_program_rewind(adk_program* p, int32_t frames);
/* The DEFERRED form of adk_program_flags, for callers that keep several steps in flight (audiodec_amd/pipeline.py, lazy_guard.py).
 * adk_program_flags_post records an event behind the launches of the step(s) just issued on `stream`; nothing waits, nothing is launched
 * (ABI 14: a program's flag word is pinned host memory its kernels report into directly; ABI 13 launched a 1-thread kernel per post).  *ticket
 * names the post (tickets count up from 0; the events of the last ADK_POST_SLOTS posts are kept).
 * adk_program_flags_poll(ticket, block): once the event has completed (block != 0: wait for it) *done = 1 and *flags = the program's word,
 * read AND cleared; else *done = 0.  ADK_ERR_STATE for a ticket that was never issued or whose slot has been reused.  With several steps of
 * one program in flight a word may already hold what a LATER step reported: a failure is attributed to the polled step or an earlier
 * one, never to a later one -- a caller that repairs by rewinding from the step a poll blames (and everything after it) is exact. */
enum { ADK_POST_SLOTS = 32 };
int adk_program_flags_post(adk_program* p, void* stream, int64_t* ticket);
int adk_program_flags_poll(adk_program* p, int64_t ticket, int32_t block, int32_t* done, int32_t* flags);
/* "Fresh" = no step since create / reset: the one state bit of a program besides its arena and cursors (the offline lowering's
 * ADK_OP_HIST_REPLICATE -- the replication pad of CausalConvTranspose1d.forward, layers/conv_layer.py:189-192 -- runs on a fresh
 * step only).  adk_program_rewind restores the value from before the rewound step -- also when it is applied several times in a row
 * (the program counts its steps since the last reset, rewinds subtracted); get / set carry the bit over to a program that takes this
 * one's place (the exact-f32 twin of the guard, also for offline programs): set(0) marks "not fresh" for good, set(1) restarts the count. */
int adk_program_get_fresh(const adk_program* p);
int adk_program_set_fresh(adk_program* p, int32_t fresh);
/* How many persistent workgroups the stream-K conv launches of this program use (multiple of 8; 0 = default = the whole
 * chip, 2 per CU).  A caller that runs several programs CONCURRENTLY on different HIP streams (software pipeline over
 * batches, bench.py) gives each a share: at 3 concurrent programs 256 measured best (210 k vs 189 k frames/s). */
int adk_program_set_workgroups(adk_program* p, int32_t workgroups);
/* Replay the steady state as HIP graphs (hipStreamBeginCapture / hipGraphLaunch): the ops between the first and the last
 * one that touch caller buffers are captured once per cursor PHASE and replayed by one hipGraphLaunch per step; the ops on
 * caller buffers (audio / codes in, waveform out) stay ordinary launches because their pointers change from call to call.
 * Ring cursors are kernel arguments, so a captured launch sequence is only valid for the cursor state it was captured in:
 * that state repeats with a short period when every ring length is a small multiple of its per-step advance
 * (max_frames * rate) -- the caller sizes the rings so by rounding `hist` up (audiodec_amd/program.py does); otherwise this
 * call fails and the program stays eager.  Only steps of exactly max_frames frames in a recognised phase, on a stream other than
 * the legacy default stream (which cannot be captured), replay; any other step (short chunk, first step after reset,
 * profiling) runs eagerly, results are identical either way.  The stream-K publish
 * flags are zeroed by a memset node at the head of each graph (their per-launch epochs are frozen by the capture). */
int adk_program_set_graph(adk_program* p, int32_t enabled);
int adk_program_graph_stats(const adk_program* p, int64_t* replays, int64_t* captures, int32_t* period);
/* ring cursors (n_rings int32), for snapshot / restore of a warmed-up state together with the arena */
int adk_program_get_cursors(const adk_program* p, int32_t* cursors, int32_t n);
int adk_program_set_cursors(adk_program* p, const int32_t* cursors, int32_t n);
/* name of the kernel op `op` runs for a `frames`-hop step (e.g. "conv_mfma<64,64>"), for profiles */
int adk_program_describe_op(adk_program* p, int32_t op, int32_t frames, char* buf, int32_t n);
/* timing aid for bench.py: per-op HIP-event durations (ms) of the LAST step when enabled; synchronises */
int adk_program_set_profiling(adk_program* p, int32_t enabled);
int adk_program_last_op_ms(adk_program* p, float* ms, int32_t n);

#ifdef __cplusplus
}
#endif
#endif /* AUDIODEC_HIP_H */
