"""Lowering of the reference's module graphs to ring-buffer launch programs.

The reference walks Python modules per call (StreamGenerator.encode/decode,
/root/reference/models/autoencoder/AudioDec.py:228-247, models/vocoder/HiFiGAN.py:268-296).  Here
each model half is lowered once, at load time, to a flat list of fused causal-conv ops over
channel-last state rings (include/audiodec_hip.h) and executed by the C++ program runner.

Lowering rules (reference semantics preserved op for op):
  * a CausalConv1d / CausalConvTranspose1d input + its pad_buffer  -> one ring with `hist` rows
    (layers/conv_layer.py:141,153-156 / :182,194-197); the ring holds the RAW signal, the consumer
    applies the activation the reference applied before the conv
  * transposed conv (K = 2*stride)  -> polyphase 2-tap conv with s*Cout GEMM rows
  * x + conv2(act(conv1(act(x))))   -> conv1 into a scratch ring, 1x1 conv with residual epilogue
    (models/autoencoder/modules/residual_unit.py:78-81)
  * MultiGroupConv1d: x.repeat(1,3,1) is never materialised: group stride 0 on the first conv's
    input and on the first residual (models/vocoder/modules/multi_fusion.py:133-141)
"""
import ctypes as C
import os

import numpy as np
import torch

from . import arch, native
from .native import (ACT_ELU, ACT_LEAKY, ACT_NONE, ACT_TANH, IMPL_AUTO, OP_CONV, OP_HIST_REPLICATE, OP_MEAN, OP_RING_WRITE,
                     ConvDesc, OpDesc, RingDesc)


# ---------------------------------------------------------------------------------------------
# weights
# ---------------------------------------------------------------------------------------------
def effective_weight(sd, spec):
    """Fold weight-norm exactly as the reference's forward pre-hook does (torch._weight_norm, dim 0)."""
    if spec.wn:
        return torch._weight_norm(sd[spec.wkey("weight_v")].float(), sd[spec.wkey("weight_g")].float(), 0)
    return sd[spec.wkey("weight")].float()


def pack_conv(w):
    """(Cout, Cin/g, K) -> [Cout][K*Cin/g]  (tap-major, channel-minor GEMM rows)."""
    co, ci, k = w.shape
    return w.permute(0, 2, 1).reshape(co, k * ci).contiguous()


def pack_convtr(w, stride):
    """ConvTranspose1d weight (Cin, Cout, 2s) -> polyphase GEMM rows [(r*Cout+co)][(j, ci)].

    y[co, t*s+r] = b[co] + sum_ci x[ci,t-1]*W[ci,co,s+r] + x[ci,t]*W[ci,co,r]   (tap 0 = older row).
    """
    ci, co, k = w.shape
    assert k == 2 * stride
    old = w[:, :, stride:].permute(2, 1, 0)     # (r, co, ci): multiplies x[t-1]
    new = w[:, :, :stride].permute(2, 1, 0)     # (r, co, ci): multiplies x[t]
    return torch.cat([old, new], dim=2).reshape(stride * co, 2 * ci).contiguous()


def pack_mfma(w_rows, groups):
    """Row-major GEMM rows [groups*cout_g][ktot] -> MFMA-fragment order (same layout as
    adk_pack_weights_mfma): [g][m-tile of 32][k-group of 8][lane 64][4], lane (i = lane & 31,
    h = lane >> 5) holding W[32*mt + i][8*kg + 4*h + 0..3]; rows beyond cout_g and the K tail
    (K padded to a multiple of 64) are zero."""
    m, ktot = w_rows.shape
    cout_g = m // groups
    assert ktot % 8 == 0
    mt32 = (cout_g + 31) // 32
    w = w_rows.reshape(groups, cout_g, ktot)
    if mt32 * 32 != cout_g:
        w = torch.cat([w, torch.zeros(groups, mt32 * 32 - cout_g, ktot)], 1)
    kpad = (ktot + 63) // 64 * 64                      # the kernel walks K in 64-deep chunks
    if kpad != ktot:
        w = torch.cat([w, torch.zeros(groups, mt32 * 32, kpad - ktot)], 2)
        ktot = kpad
    w = w.reshape(groups, mt32, 32, ktot // 8, 2, 4)        # (g, mt, i, kg, h, e)
    return w.permute(0, 1, 3, 4, 2, 5).contiguous().reshape(-1)   # (g, mt, kg, h, i, e): lane = h*32 + i


def pack_split16(w_rows, groups):
    """Row-major GEMM rows [groups*cout_g][ktot] -> the split-f16 fragment order of adk_pack_weights_split16:
    [g][m-tile of 32][16-k chunk][hi | lo][lane 64][8 x f16], lane (i = lane & 31, h = lane >> 5) holding
    W[32*mt + i][16*chunk + 8*h + 0..7]; hi = f16(W), lo = f16((W - hi) * 2048); rows beyond cout_g and the K
    tail (K padded to a multiple of 64) are zero.  Returned as a float32 view (two halfs per float) so it can
    live in the weight blob."""
    m, ktot = w_rows.shape
    cout_g = m // groups
    assert ktot % 16 == 0
    mt32 = (cout_g + 31) // 32
    w = w_rows.reshape(groups, cout_g, ktot).float()
    if mt32 * 32 != cout_g:
        w = torch.cat([w, torch.zeros(groups, mt32 * 32 - cout_g, ktot)], 1)
    kpad = (ktot + 63) // 64 * 64                      # same 64-deep chunking as pack_mfma
    if kpad != ktot:
        w = torch.cat([w, torch.zeros(groups, mt32 * 32, kpad - ktot)], 2)
        ktot = kpad
    if float(w.abs().max()) > 65504.0:
        raise ValueError("split-f16 weights must be below 65504 in magnitude")
    hi = w.half()
    lo = ((w - hi.float()) * 2048.0).half()
    both = torch.stack([hi, lo], 0).reshape(2, groups, mt32, 32, ktot // 16, 2, 8)     # (p, g, mt, i, ch, h, j)
    out = both.permute(1, 2, 4, 0, 5, 3, 6).contiguous().reshape(-1)                   # (g, mt, ch, p, h, i, j)
    return out.view(torch.float32)


def split16_eligible(cin_g, cout_g, groups):
    """Layers that have a split-f16 kernel: everything the matrix-core kernels take (conv_sk16 / conv_rl16)."""
    return mfma_eligible(cin_g, cout_g, groups) and cin_g % 32 == 0


def mfma_eligible(cin_g, cout_g, groups):
    return cin_g % 32 == 0 and cout_g % 4 == 0 and groups * cout_g >= 32


FUSE_RES_UNITS = os.environ.get("ADK_FUSE", "1") != "0"      # ADK_FUSE=0: every residual unit as two launches (A/B, cross-checks)
FUSE_CHAINS = os.environ.get("ADK_CHAIN", "1") != "0"        # ADK_CHAIN=0: residual chains (a block's three units) op by op, not as one launch
SHADOW_RINGS = os.environ.get("ADK_SHADOW", "1") != "0"      # ADK_SHADOW=0: no shadow rings (every stream-K conv activates + splits what it stages; A/B, cross-checks)
SHADOW_MIN_CH = 128                                           # channels per group from which a conv is always a stream-K launch (the rows / chain kernels take 32 / 64, chains 128)


class Blob:
    """Packed fp32 weights; every tensor starts on a 16-byte boundary."""

    def __init__(self):
        self.parts, self.n = [], 0

    def add(self, t):
        t = t.detach().float().contiguous().reshape(-1)
        off = self.n
        pad = (-t.numel()) % 4
        self.parts.append(t)
        if pad:
            self.parts.append(torch.zeros(pad))
        self.n += t.numel() + pad
        return off

    def tensor(self):
        return torch.cat(self.parts) if self.parts else torch.zeros(4)


# ---------------------------------------------------------------------------------------------
# program builder
# ---------------------------------------------------------------------------------------------
class Builder:
    def __init__(self, sd, specs, offline=False, split16=False):
        self.sd = sd
        self.split16 = split16          # opt-in: f16 hi/lo split operands on the f16 matrix cores where a kernel exists
        self.offline = offline          # lower Generator.forward (file-level drivers) instead of the streaming inference
        self.specs = arch.by_name(specs)
        self.blob = Blob()
        self.rings, self.ops, self.op_names = [], [], []
        self.scratch = {}
        self.flops_per_frame = 0
        self._auto = []                 # per op: lowered with impl = AUTO (the shape, not the arithmetic of this builder, decides about shadow rings)
        self._shadows_done = False
        self.shadow_of = {}             # ring id -> id of its shadow ring

    def ring(self, channels, hist, rate, external=-1):
        self.rings.append(dict(channels=channels, hist=hist, rate=rate, external=external))
        return len(self.rings) - 1

    def scratch_ring(self, channels, rate, tag=""):
        key = (channels, rate, tag)
        if key not in self.scratch:
            self.scratch[key] = self.ring(channels, 0, rate)
        return self.scratch[key]

    def need_hist(self, ring_id, hist):
        r = self.rings[ring_id]
        assert r["external"] < 0 or hist == 0
        r["hist"] = max(r["hist"], hist)

    def ring_write(self, out_ring, ext_src, mean=None, scale=None):
        op = OpDesc()
        op.kind = OP_RING_WRITE
        op.in_ring, op.out_ring, op.res_ring = -1, out_ring, -1
        op.ext_src = ext_src
        op.mean_off = self.blob.add(mean) if mean is not None else -1
        op.scale_off = self.blob.add(scale) if scale is not None else -1
        op.w_off, op.wf_off, op.b_off = -1, -1, -1
        op.rate_out = self.rings[out_ring]["rate"]
        self.ops.append(op)
        self._auto.append(False)
        self.op_names.append("ring_write")

    def mean(self, src_rings, out_ring):
        """out = (((s0 + s1) + s2) ...) / n   (MultiReceptiveField, multi_fusion.py:73-79)"""
        op = OpDesc()
        op.kind = OP_MEAN
        op.in_ring, op.out_ring, op.res_ring = -1, out_ring, -1
        op.w_off, op.wf_off, op.b_off, op.mean_off, op.scale_off = -1, -1, -1, -1, -1
        op.ext_src = -1
        op.n_mean = len(src_rings)
        for k, r in enumerate(src_rings):
            op.mean_rings[k] = r
        op.rate_out = self.rings[out_ring]["rate"]
        self.ops.append(op)
        self._auto.append(False)
        self.op_names.append("mean")

    def hist_replicate(self, ring):
        """First step after a reset: history rows of `ring` <- its first new row (the ReplicationPad1d of
        CausalConvTranspose1d.forward, layers/conv_layer.py:189-192)."""
        op = OpDesc()
        op.kind = OP_HIST_REPLICATE
        op.in_ring, op.out_ring, op.res_ring = ring, -1, -1
        op.w_off, op.wf_off, op.b_off, op.mean_off, op.scale_off = -1, -1, -1, -1, -1
        op.ext_src = -1
        op.rate_out = self.rings[ring]["rate"]
        self.ops.append(op)
        self._auto.append(False)
        self.op_names.append("hist_replicate")

    def conv(self, name, in_ring, out_ring, act_in=ACT_NONE, slope=0.0, act_out=ACT_NONE, res_ring=-1,
             in_group_stride=None, res_group_stride=None, impl=IMPL_AUTO, fuse_next=False):
        s = self.specs[name]
        w = effective_weight(self.sd, s)
        bias = self.sd[s.wkey("bias")].float() if s.bias else None
        d = ConvDesc()
        rin, rout = self.rings[in_ring], self.rings[out_ring]
        if s.kind == "convT":
            packed = pack_convtr(w, s.stride)
            d.cin_g, d.cout_g, d.groups = s.cin, s.stride * s.cout, 1
            d.taps, d.stride, d.dilation, d.hist = 2, 1, 1, 1
            d.up, d.cout_real = s.stride, s.cout
            if bias is not None:
                bias = bias.repeat(s.stride)
            rate_out = rin["rate"]
            assert rout["rate"] == rin["rate"] * s.stride
        else:
            packed = pack_conv(w)
            d.cin_g, d.cout_g, d.groups = s.cin // s.groups, s.cout // s.groups, s.groups
            d.taps, d.stride, d.dilation, d.hist = s.k, s.stride, s.dilation, s.pad
            d.up, d.cout_real = 1, s.cout
            assert rin["rate"] % s.stride == 0 and rout["rate"] == rin["rate"] // s.stride, name
            rate_out = rout["rate"]
        d.in_group_stride = d.cin_g if in_group_stride is None else in_group_stride
        d.res_group_stride = d.cout_g if res_group_stride is None else res_group_stride
        d.act_in, d.act_in_slope, d.act_out = act_in, slope, act_out
        self.need_hist(in_ring, d.hist)
        if s.kind == "convT" and self.offline:
            self.hist_replicate(in_ring)
        op = OpDesc()
        op.kind = OP_CONV
        op.in_ring, op.out_ring, op.res_ring = in_ring, out_ring, res_ring
        op.in_ch_off = op.out_ch_off = op.res_ch_off = 0
        op.rate_out = rate_out
        op.conv = d
        auto = impl == IMPL_AUTO
        if self.split16 and impl == IMPL_AUTO and split16_eligible(d.cin_g, d.cout_g, d.groups):
            impl = native.IMPL_SPLIT16
            op.w_off, op.wf_off = -1, self.blob.add(pack_split16(packed, d.groups))
        elif mfma_eligible(d.cin_g, d.cout_g, d.groups) and impl != native.IMPL_DIRECT:
            op.w_off, op.wf_off = -1, self.blob.add(pack_mfma(packed, d.groups))
        else:
            op.w_off, op.wf_off = self.blob.add(packed), -1
        op.b_off = self.blob.add(bias) if bias is not None else -1
        op.mean_off = op.scale_off = -1
        op.ext_src = -1
        op.impl = impl
        op.fuse_next = 1 if fuse_next else 0
        op.chain = 0
        self.ops.append(op)
        self._auto.append(auto)
        self.op_names.append(name)
        self.flops_per_frame += 2 * packed.numel() * rate_out
        return op

    def assign_shadows(self):
        """Give the rings between two stream-K convs a SHADOW ring (adk_op_desc.in_shadow / out_shadow): the producer's epilogue stores,
        beside every 4 floats, their split-f16 operand form [4 x f16 hi][4 x f16 lo] of act(x); the consumer stages those 16 bytes as
        they are.  Without it the consumer re-applies the activation and the split to every element it stages -- once per tap and per
        64-row tile of output channels: 44 times per element in the 256-channel grouped K11 convs of a v1 vocoder's first stage.

        A ring qualifies when EVERY op that writes it and at least one conv that reads it are certain to run on the stream-K kernel in
        the split-f16 lowering: AUTO-lowered convs with >= SHADOW_MIN_CH channels per group (the rows kernels take 32 / 64), outside a
        chain the chain kernel takes (<= 128 channels) and outside a fusable pair; the qualifying readers must agree on the input
        activation.  The decision looks at shapes only, so that the exact-f32 twin of a program (HipProgram.demote) lays out the same
        arena: it allocates the shadow rings and leaves them alone.  Not for offline programs (their history replicate op would have to
        replicate shadows too).  Called once, by HipProgram, after the last op was added."""
        if self._shadows_done:
            return
        self._shadows_done = True
        if self.offline or not SHADOW_RINGS:
            return
        n = len(self.ops)
        in_chain, in_pair = set(), set()
        for i, op in enumerate(self.ops):
            if op.kind != OP_CONV:
                continue
            if op.chain >= 2:
                in_chain.update(range(i, i + op.chain))
            if op.fuse_next and i + 1 < n and self.ops[i + 1].kind == OP_CONV and self.ops[i + 1].conv.cin_g in (32, 64):
                in_pair.update((i, i + 1))      # residual unit on the rows kernel / conv_out + the 64-channel transposed conv (conv_ou16)

        def streamk(i):
            op = self.ops[i]
            if op.kind != OP_CONV or not self._auto[i] or i in in_pair:
                return False
            d = op.conv
            if not split16_eligible(d.cin_g, d.cout_g, d.groups) or d.cin_g < SHADOW_MIN_CH:
                return False
            return not (i in in_chain and d.cin_g <= 128)

        for rid, r in enumerate(list(self.rings)):
            if r["external"] >= 0:
                continue
            writers = [i for i, op in enumerate(self.ops) if op.out_ring == rid and op.kind != OP_HIST_REPLICATE]
            readers = [i for i, op in enumerate(self.ops) if op.kind == OP_CONV and op.in_ring == rid]
            if not writers or not all(streamk(i) for i in writers):
                continue
            if any(self.ops[i].out_ch_off != 0 or self.ops[i].conv.cout_real != r["channels"] for i in writers):
                continue                        # (a shadow is complete only if its writers write whole rows)
            sk_readers = [i for i in readers if streamk(i)]
            acts = {(self.ops[i].conv.act_in, float(self.ops[i].conv.act_in_slope)) for i in sk_readers}
            if not sk_readers or len(acts) != 1:
                continue
            act, slope = next(iter(acts))
            if act not in (ACT_NONE, ACT_ELU, ACT_LEAKY):
                continue
            sh = self.ring(r["channels"], r["hist"], r["rate"])
            self.rings[sh]["shadow_of"] = rid
            self.shadow_of[rid] = sh
            if not self.split16:
                continue                        # exact-f32 lowering: same arena layout, nobody touches the shadow
            for i in writers:
                op = self.ops[i]
                op.out_shadow, op.shadow_act, op.shadow_slope = sh + 1, act, slope
                op.impl = native.IMPL_SPLIT16_SK
            for i in sk_readers:
                self.ops[i].in_shadow = sh + 1


def _act_of(params, default="ELU"):
    name = params.get("nonlinear_activation", default)
    ap = params.get("nonlinear_activation_params", {}) or {}
    if name == "ELU":
        if float(ap.get("alpha", 1.0)) != 1.0:
            raise NotImplementedError("ELU alpha != 1 is not supported")
        return ACT_ELU, 0.0
    if name == "LeakyReLU":
        return ACT_LEAKY, float(ap.get("negative_slope", 0.01))
    raise NotImplementedError(f"Activation {name} is not supported!")


def _res_units(b, pre, x_ring, c, rate, act, slope, out_ring_of_last):
    """3x CausalResidualUnit.inference; returns the ring holding the block output."""
    first = None
    for j in range(3):
        h = b.scratch_ring(c, rate, "h")
        # h is read by conv2 only: the runner may run the unit as one kernel (adk_op_desc.fuse_next)
        op = b.conv(f"{pre}.res_units.{j}.conv1", x_ring, h, act, slope, fuse_next=FUSE_RES_UNITS)
        first = first or op
        nxt = out_ring_of_last if j == 2 else b.ring(c, 0, rate)
        b.conv(f"{pre}.res_units.{j}.conv2", h, nxt, act, slope, res_ring=x_ring)
        x_ring = nxt
    # the three units read / write rings nobody else touches in between: the runner may run all six convs as one launch
    first.chain = 6 if FUSE_CHAINS else 0
    return x_ring


def build_encoder(sd, p, split16=False):
    """Encoder.encode + Projector.encode (encoder.py:137-142, projector.py:52-54).  ext: [x, z]."""
    specs = arch.autoencoder_encoder_convs(p)
    b = Builder(sd, specs, split16=split16)
    act, slope = _act_of(p)
    hop = arch.hop_length(p)
    in_ch = p.get("input_channels", 1)
    activate = p.get("codec", "audiodec") == "activate_audiodec"
    ch, ratios, strides = p.get("encode_channels", 32), p.get("enc_ratios", (2, 4, 8, 16)), p.get("enc_strides", (3, 4, 5, 5))
    rx = b.ring(in_ch, 0, hop)
    b.ring_write(rx, 0)
    rate, c = hop, ch
    cur = b.ring(c, 0, rate)
    b.conv("encoder.conv", rx, cur)
    for i, s in enumerate(strides):
        pre = f"encoder.conv_blocks.{i}"
        last = b.ring(c, 0, rate)
        cur = _res_units(b, pre, cur, c, rate, act, slope, last)
        c2, rate2 = ch * ratios[i], rate // s
        nxt = b.ring(c2, 0, rate2)
        b.conv(f"{pre}.conv", cur, nxt)
        cur, c, rate = nxt, c2, rate2
    z = b.ring(p.get("code_dim", 64), 0, rate, external=1)
    # ActivateEncoder applies the activation to the encoder output (encoder.py:171-175)
    b.conv("projector.project", cur, z, act if activate else ACT_NONE, slope)
    return b


def build_sym_decoder(sd, p, offline=False, split16=False):
    """Decoder.decode / ActivateDecoder.decode (decoder.py:142-148, 203-214), or with offline=True
    Decoder.forward (:136-140: same layers, replication pad in front of the transposed convs).  ext: [zq, y]."""
    specs = arch.autoencoder_decoder_convs(p)
    b = Builder(sd, specs, offline, split16)
    act, slope = _act_of(p)
    activate = p.get("codec", "audiodec") == "activate_audiodec"
    ch, ratios, strides = p.get("decode_channels", 32), p.get("dec_ratios", (16, 8, 4, 2)), p.get("dec_strides", (5, 5, 4, 3))
    rate = 1
    rz = b.ring(p.get("code_dim", 64), 0, rate)
    b.ring_write(rz, 0)
    cur = b.ring(ch * ratios[0], 0, rate)
    b.conv("decoder.conv1", rz, cur)
    for i, s in enumerate(strides):
        cout = ch * ratios[i + 1] if i < len(ratios) - 1 else ch
        pre = f"decoder.conv_blocks.{i}.1" if activate else f"decoder.conv_blocks.{i}"
        rate *= s
        up = b.ring(cout, 0, rate)
        b.conv(f"{pre}.conv", cur, up, act if activate else ACT_NONE, slope)
        last = b.ring(cout, 0, rate)
        cur = _res_units(b, pre, up, cout, rate, act, slope, last)
    y = b.ring(p.get("output_channels", 1), 0, rate, external=1)
    b.conv("decoder.conv2", cur, y, act if activate else ACT_NONE, slope, ACT_TANH if activate else ACT_NONE)
    return b


def hifigan_stage_boundary(p, split_at):
    """(channels, rows per frame) of the tensor between upsample stage split_at-1 and split_at."""
    ch = p.get("channels", 512)
    rate = 1
    for s_ in p["upsample_scales"][:split_at]:
        rate *= s_
    return ch // (2 ** split_at), rate


def build_hifigan(sd, p, offline=False, split16=False, part=None, split_at=2):
    """See below; `split_at` may be an int (one cut) or a sorted list of cut points (len+1 programs)."""
    return _build_hifigan(sd, p, offline, split16, part, [split_at] if isinstance(split_at, int) else list(split_at))


def _build_hifigan(sd, p, offline, split16, part, cuts):
    """HiFiGAN StreamGenerator.decode (HiFiGAN.py:268-296), or with offline=True Generator.forward
    (:141-161).  ext: [zq, y].

    part = k lowers program k of the path cut in front of the upsample stages listed in `cuts` (ext: [in, out] of that
    program; hand-over tensors are (B, frames*rate, channels) channel-last): consecutive batches can then be
    software-pipelined over HIP streams, program k of batch i+1 under program k+1 of batch i (bench.py).  Same ops,
    same arithmetic."""
    specs = arch.hifigan_convs(p)
    b = Builder(sd, specs, offline, split16)
    act, slope = _act_of(p, "LeakyReLU")
    multigroup = arch.hifigan_is_multigroup(p)
    groups = p.get("groups", 1)
    ch = p.get("channels", 512)
    addl = p.get("use_additional_convs", True)       # False: x + convs1(act(x)), no second conv (residual_block.py:100-105)
    n_up = len(p["upsample_scales"])
    if part is not None:
        if not cuts or sorted(set(cuts)) != list(cuts) or cuts[0] <= 0 or cuts[-1] >= n_up or not 0 <= part <= len(cuts):
            raise ValueError("cut points must be distinct, ascending and lie between two upsample stages")
    bounds = [0] + (list(cuts) if part is not None else []) + [n_up]
    first = bounds[part] if part is not None else 0
    last = bounds[part + 1] if part is not None else n_up
    split_at = first
    rate = 1
    if first == 0:
        rz = b.ring(p["in_channels"], 0, rate)
        norm = "mean" in sd
        b.ring_write(rz, 0, sd["mean"] if norm else None, sd["scale"] if norm else None)
        cur = b.ring(ch, 0, rate)
        b.conv("input_conv", rz, cur)
    else:
        c_mid, rate = hifigan_stage_boundary(p, split_at)
        cur = b.ring(c_mid, 0, rate)
        b.ring_write(cur, 0)
    c = ch
    for i in range(first, last):
        s = p["upsample_scales"][i]
        c = ch // (2 ** (i + 1))
        rate *= s
        x0 = b.ring(c, 0, rate)                                 # block input, un-repeated
        b.conv(f"upsamples.{i}", cur, x0, act, slope)           # upsamples[i].inference(act(c))
        cur_internal = not (part is not None and last < n_up and i == last - 1)
        cur = b.ring(c, 0, rate, external=-1 if cur_internal else 1)
        if multigroup:
            # MultiGroupConv1d.inference (multi_fusion.py:133-141): x.repeat(1, groups, 1) is never
            # materialised -- the first conv and the first residual read the same C channels per group
            x, gs_in, gs_res = x0, 0, 0
            n_units = len(p["resblock_dilations"][0])
            for j in range(n_units):
                if addl:
                    xt = b.ring(c * groups, 0, rate)
                    op = b.conv(f"blocks.{i}.convs1.{j}", x, xt, act, slope, in_group_stride=gs_in)
                    if j == 0 and FUSE_CHAINS and 2 * n_units <= 8:
                        op.chain = 2 * n_units                   # HiFiGANResidualBlock.inference as one launch (residual_block.py:99-105)
                    nx = b.ring(c * groups, 0, rate)
                    b.conv(f"blocks.{i}.convs2.{j}", xt, nx, act, slope, res_ring=x, res_group_stride=gs_res)
                else:
                    nx = b.ring(c * groups, 0, rate)
                    b.conv(f"blocks.{i}.convs1.{j}", x, nx, act, slope, in_group_stride=gs_in, res_ring=x, res_group_stride=gs_res)
                x, gs_in, gs_res = nx, None, None
            # the 1x1 conv_out feeds only the next stage's activation + transposed conv: the runner may run the two as one
            # streaming launch (csrc/conv_ou16.hip; the 64-channel tensor between them then never exists in memory)
            nxt_is_up = i + 1 < last and cur_internal and not offline and FUSE_RES_UNITS
            # ... and the last one only the output conv: csrc/conv_oc16.hip (of the 32-channel tensor only the last 6 steps are stored)
            nxt_is_out = i + 1 == n_up and last == n_up and not offline and FUSE_RES_UNITS
            b.conv(f"blocks.{i}.conv_out", x, cur, fuse_next=nxt_is_up or nxt_is_out)
        else:
            # MultiReceptiveField.inference (multi_fusion.py:73-79): mean of the residual blocks, all fed by x0
            outs = []
            for bi, dil in enumerate(p["resblock_dilations"]):
                x = x0
                for j in range(len(dil)):
                    if addl:
                        xt = b.ring(c, 0, rate)
                        op = b.conv(f"blocks.{i}.blocks.{bi}.convs1.{j}", x, xt, act, slope)
                        if j == 0 and FUSE_CHAINS and 2 * len(dil) <= 8:
                            op.chain = 2 * len(dil)
                        nx = b.ring(c, 0, rate)
                        b.conv(f"blocks.{i}.blocks.{bi}.convs2.{j}", xt, nx, act, slope, res_ring=x)
                    else:
                        nx = b.ring(c, 0, rate)
                        b.conv(f"blocks.{i}.blocks.{bi}.convs1.{j}", x, nx, act, slope, res_ring=x)
                    x = nx
                outs.append(x)
            b.mean(outs, cur)
    if last == n_up:
        y = b.ring(p.get("out_channels", 1), 0, rate, external=1)
        # activation_output1 = nn.LeakyReLU() default slope 0.01 (HiFiGAN.py:116); tanh after (:117)
        b.conv("output_conv", cur, y, ACT_LEAKY, 0.01, ACT_TANH)
    return b


# ---------------------------------------------------------------------------------------------
# runtime wrapper
# ---------------------------------------------------------------------------------------------
_NICE_PERIODS = (1, 2, 3, 4, 6, 12, 24, 48)


def graph_ring_hist(hist, rate, max_frames, extra_steps=0):
    """History a ring must keep so that its cursor returns to the same row after a SMALL number of full-size steps: the ring
    length hist + (1 + extra_steps) * max_frames * rate becomes a multiple P * (max_frames*rate) with P from _NICE_PERIODS (all
    divide 48, so the periods of all rings of a program have a small common multiple).  That is what lets adk_program_set_graph
    capture one launch sequence per cursor phase: ring cursors are kernel arguments.  extra_steps: the rewind_depth rows the ring
    carries beyond its history and one step (adk_ring_desc.extra_rows)."""
    adv = max_frames * rate
    need = -(-(hist + adv) // adv) + extra_steps
    p = next((q for q in _NICE_PERIODS if q >= need), need)
    return (p - 1 - extra_steps) * adv


class HipProgram:
    """One model half on one HIP device for `batch` streams (C++ adk_program + arena + weights)."""

    def __init__(self, builder, batch, max_frames, device, graph=False, twin=None, rewind_depth=0):
        """rewind_depth = k: every arena ring gets k * max_frames * rate rows more than its consumers need, so that rewind() can be
        applied up to k + 1 times in a row -- the history of the step k steps back is still there (deferred guard, pipeline.py)."""
        self.dev = native.require_gpu(device)
        self.lib = native.lib()
        self.batch, self.max_frames = int(batch), int(max_frames)
        self.split16, self.offline = bool(builder.split16), bool(builder.offline)
        self.twin_builder = twin          # () -> Builder of the same lowering with the exact-f32 kernels (demote())
        self.rewind_depth = 0 if builder.offline else max(0, int(rewind_depth))
        self.graph_requested = bool(graph)
        self.demoted = False
        self.workgroups = 0
        builder.assign_shadows()
        if graph:
            for r in builder.rings:
                if r["external"] < 0:
                    r["hist"] = graph_ring_hist(r["hist"], r["rate"], self.max_frames, self.rewind_depth)
        self.op_names = list(builder.op_names)
        self.flops_per_frame = builder.flops_per_frame
        self.n_ops, self.n_rings = len(builder.ops), len(builder.rings)
        off = 0
        self.ring_rows, self.ring_meta = [], []
        rings = (RingDesc * self.n_rings)()
        for i, r in enumerate(builder.rings):
            extra = self.rewind_depth * self.max_frames * r["rate"] if r["external"] < 0 else 0
            rows = r["hist"] + self.max_frames * r["rate"] + extra if r["external"] < 0 else 0
            rings[i].channels, rings[i].hist, rings[i].rate, rings[i].external = r["channels"], r["hist"], r["rate"], r["external"]
            rings[i].extra_rows = extra
            rings[i].arena_off = off if r["external"] < 0 else 0
            self.ring_rows.append(rows)
            self.ring_meta.append(dict(r, rows=rows, arena_off=off))
            if r["external"] < 0:
                off += self.batch * rows * r["channels"]
                off += (-off) % 4
        self.arena_floats = max(off, 4)
        # (shadow rings are derived state -- the split-f16 form of rows their rings hold -- and are not counted)
        self.state_floats_per_stream = sum(r["hist"] * r["channels"] for r in builder.rings if r["external"] < 0 and "shadow_of" not in r)
        self.weights = builder.blob.tensor().to(self.dev)
        self.weight_floats = self.weights.numel()
        self.arena = torch.zeros(self.arena_floats, dtype=torch.float32, device=self.dev)
        ops = (OpDesc * self.n_ops)(*builder.ops)
        self._ops = ops
        h = C.c_void_p()
        native.check(self.lib.adk_program_create(ops, self.n_ops, rings, self.n_rings, self.batch, self.max_frames,
                                                 C.c_void_p(self.weights.data_ptr()), self.weight_floats,
                                                 C.c_void_p(self.arena.data_ptr()), self.arena_floats, C.byref(h)),
                     "adk_program_create")
        self.h = h
        self._ext = (C.c_void_p * 8)()
        self.graph = False
        if graph:
            rc = self.lib.adk_program_set_graph(self.h, 1)
            self.graph = rc == native.ADK_OK          # no short cursor period / nothing to capture: the program stays eager


    def __del__(self):
        try:
            if getattr(self, "h", None):
                self.lib.adk_program_destroy(self.h)
                self.h = None
        except Exception:
            pass

    def step(self, frames, ext, replay=False):
        """replay=True (adk_program_step_ex, ADK_STEP_REPLAY): the step is being repeated after rewind(); the ring writes are skipped
        -- the caller's input rows are still in their rings -- and the entries of `ext` they alone read may be None."""
        for i, t in enumerate(ext):
            self._ext[i] = t.data_ptr() if t is not None else None
        if replay:
            native.check(self.lib.adk_program_step_ex(self.h, int(frames), self._ext, len(ext), native.current_stream(self.dev), native.STEP_REPLAY),
                         "adk_program_step_ex")
        else:
            native.check(self.lib.adk_program_step(self.h, int(frames), self._ext, len(ext), native.current_stream(self.dev)),
                         "adk_program_step")

    def post_flags(self):
        """Deferred check (adk_program_flags_post): behind what was just issued on the current HIP stream, store-and-clear this
        program's flag word into a pinned host word; nothing waits.  Returns the ticket for poll_flags."""
        t = C.c_int64(0)
        native.check(self.lib.adk_program_flags_post(self.h, native.current_stream(self.dev), C.byref(t)), "adk_program_flags_post")
        return int(t.value)

    def poll_flags(self, ticket, block=False):
        """(done, flags) of a post: done is False while the post has not completed on the device (block=True waits for it)."""
        d, f = C.c_int32(0), C.c_int32(0)
        native.check(self.lib.adk_program_flags_poll(self.h, int(ticket), 1 if block else 0, C.byref(d), C.byref(f)), "adk_program_flags_poll")
        return bool(d.value), int(f.value)

    def reset(self):
        native.check(self.lib.adk_program_reset(self.h, native.current_stream(self.dev)), "adk_program_reset")

    def flags(self):
        """Wait for the current HIP stream, return this program's sticky device flag word and clear it (adk_program_flags)."""
        v = C.c_int32(0)
        native.check(self.lib.adk_program_flags(self.h, native.current_stream(self.dev), C.byref(v)), "adk_program_flags")
        return int(v.value)

    def rewind(self, frames):
        """Ring cursors back by the `frames` hops of the step just taken (adk_program_rewind): the step can be repeated."""
        native.check(self.lib.adk_program_rewind(self.h, int(frames)), "adk_program_rewind")

    def demote(self):
        """The exact-f32 twin of this program takes over IN PLACE (every reference to this object stays valid): same ops,
        same rings, same arena layout, weights packed for the f32 kernels; the state (arena + cursors) is carried across.
        Used when a split-f16 step reports an operand beyond the f16 range: rewind(), demote(), repeat the step."""
        if self.twin_builder is None or not self.split16 or self.demoted:
            raise native.NativeError("this program has no exact-f32 twin to fall back to")
        twin = HipProgram(self.twin_builder(), self.batch, self.max_frames, self.dev, graph=self.graph_requested, rewind_depth=self.rewind_depth)
        if twin.arena_floats != self.arena_floats or twin.n_rings != self.n_rings or twin.ring_rows != self.ring_rows:
            raise native.NativeError("the exact-f32 twin has a different state layout")
        twin.arena.copy_(self.arena)
        twin.set_cursors(self.cursors())
        # "fresh" (no step since reset) decides whether the offline lowering's history replicate runs: the twin repeats the step
        # in the state this program was in before it (adk_program_rewind put the bit back)
        native.check(self.lib.adk_program_set_fresh(twin.h, self.lib.adk_program_get_fresh(self.h)), "adk_program_set_fresh")
        if self.workgroups:
            twin.set_workgroups(self.workgroups)
        mine = dict(self.__dict__)
        for k in ("h", "weights", "weight_floats", "arena", "_ops", "op_names", "n_ops", "ring_meta", "ring_rows", "flops_per_frame",
                  "graph", "state_floats_per_stream", "arena_floats", "n_rings", "_ext"):
            self.__dict__[k] = twin.__dict__[k]
            twin.__dict__[k] = mine[k]                 # the split-f16 program is destroyed with `twin`
        self.split16, self.demoted = False, True

    def cursors(self):
        c = (C.c_int32 * self.n_rings)()
        native.check(self.lib.adk_program_get_cursors(self.h, c, self.n_rings), "adk_program_get_cursors")
        return list(c)

    def set_cursors(self, cur):
        c = (C.c_int32 * self.n_rings)(*cur)
        native.check(self.lib.adk_program_set_cursors(self.h, c, self.n_rings), "adk_program_set_cursors")

    def snapshot(self):
        return self.arena.clone(), self.cursors()

    def restore(self, snap):
        self.arena.copy_(snap[0])
        self.set_cursors(snap[1])

    def describe_op(self, op, frames):
        buf = C.create_string_buffer(64)
        native.check(self.lib.adk_program_describe_op(self.h, op, frames, buf, 64), "adk_program_describe_op")
        return buf.value.decode()

    # ---- per-stream state (multi-stream serving: one stream joins / leaves / is reset, the others run on) ----
    def _ring_view(self, i):
        m = self.ring_meta[i]
        n = self.batch * m["rows"] * m["channels"]
        return self.arena[m["arena_off"]:m["arena_off"] + n].view(self.batch, m["rows"], m["channels"])

    def capture_stream_state(self, b=0):
        """History rows of stream b, oldest first, per ring: what the reference calls the pad_buffers.
        Cursors only ever advance by whole frames, so the rows in front of the cursor keep their phase
        and the capture can be restored later at a different cursor."""
        cur = self.cursors()
        out = []
        for i, m in enumerate(self.ring_meta):
            if m["external"] >= 0 or m["hist"] == 0:
                out.append(None)
                continue
            rows = (cur[i] - m["hist"] + torch.arange(m["hist"], device=self.dev)) % m["rows"]
            v = self._ring_view(i)
            if "shadow_of" in m:
                v = v.view(torch.int32)         # f16 pairs: raw bits, never arithmetic
            out.append(v[b, rows].clone())
        return out

    def restore_stream_state(self, b, state):
        """Overwrite stream b's history with a captured state (state=None: zeros = reset_buffer for that stream)."""
        cur = self.cursors()
        for i, m in enumerate(self.ring_meta):
            if m["external"] >= 0 or m["hist"] == 0:
                continue
            rows = (cur[i] - m["hist"] + torch.arange(m["hist"], device=self.dev)) % m["rows"]
            v = self._ring_view(i)
            if "shadow_of" in m:
                v = v.view(torch.int32)         # (split(act(0)) = 0 for ELU / LeakyReLU / none: a zeroed shadow matches a zeroed ring)
            if state is None:
                v[b, rows] = 0
            else:
                v[b, rows] = state[i]

    def graph_stats(self):
        """(graph replays, graphs captured, cursor period) -- (0, 0, 0) for an eager program."""
        r, c, per = C.c_int64(0), C.c_int64(0), C.c_int32(0)
        native.check(self.lib.adk_program_graph_stats(self.h, C.byref(r), C.byref(c), C.byref(per)), "adk_program_graph_stats")
        return int(r.value), int(c.value), int(per.value)

    def set_workgroups(self, workgroups):
        """Persistent workgroups per stream-K launch (0 = whole chip); see adk_program_set_workgroups."""
        self.workgroups = int(workgroups)
        native.check(self.lib.adk_program_set_workgroups(self.h, int(workgroups)), "adk_program_set_workgroups")

    def set_profiling(self, on):
        native.check(self.lib.adk_program_set_profiling(self.h, 1 if on else 0), "adk_program_set_profiling")

    def last_op_ms(self):
        ms = (C.c_float * self.n_ops)()
        native.check(self.lib.adk_program_last_op_ms(self.h, ms, self.n_ops), "adk_program_last_op_ms")
        return list(ms)
