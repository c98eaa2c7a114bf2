"""CPU ORACLE for the AudioDec streaming hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` may
import this file.  The product path (``audiodec_amd``) never does: it runs the hand-written HIP
kernels in ``audiodec_amd/csrc`` through the C ABI of ``include/audiodec_hip.h`` and raises if
the shared library is missing.

What this is: a restatement, function by function, of the reference's streaming arithmetic.
The reference (facebookresearch/AudioDec, ~5.6 kLoC of Python) keeps all of its arithmetic in
**PyTorch itself** (``torch.nn.functional.conv1d / conv_transpose1d / embedding``, ``@``, ``max``;
SURVEY.md section 8c) -- PyTorch 2.10.0 is the pinned third-party dependency and is present on the
GPU box, so this oracle calls the very same ATen CPU kernels in fp32, in the same order, with the
same operand shapes.  Each function cites the reference file:line it follows.

Parity pin: ``tests/golden/make_golden.py`` runs the *unmodified* reference (imported from
/root/reference, CPU) and this oracle on the same seeded synthetic checkpoints and inputs; the
two agree bit-for-bit in the build container (asserted by that script), and the reference
outputs are committed as fixtures under ``tests/golden/*.npz``.  ``tests/test_oracle_golden.py``
re-checks oracle vs fixtures wherever the suite runs.
"""
import math
import numpy as np
import torch
import torch.nn.functional as F


# ----------------------------------------------------------------------------------------------
# layers/conv_layer.py
# ----------------------------------------------------------------------------------------------
def causal_conv1d_inference(x, pad_buffer, weight, bias, stride=1, dilation=1, groups=1):
    """CausalConv1d.inference (layers/conv_layer.py:153-156).

    x (B, Cin, L); pad_buffer (B, Cin, P) with P = (K-1)*dilation (:141).  Returns (y, new_pad).
    """
    P = pad_buffer.shape[-1]
    x = torch.cat((pad_buffer, x), -1)                       # :154
    new_pad = x[:, :, x.shape[-1] - P:] if P > 0 else x[:, :, :0]   # :155  (x[:, :, -P:])
    y = F.conv1d(x, weight, bias, stride=stride, padding=0, dilation=dilation, groups=groups)  # :156
    return y, new_pad


def causal_conv1d_forward(x, weight, bias, stride=1, dilation=1, groups=1):
    """CausalConv1d.forward: zero left-pad (layers/conv_layer.py:148-151)."""
    P = (weight.shape[-1] - 1) * dilation
    x = F.pad(x, (P, 0), value=0.0)
    return F.conv1d(x, weight, bias, stride=stride, padding=0, dilation=dilation, groups=groups)


def causal_convtr1d_inference(x, pad_buffer, weight, bias, stride):
    """CausalConvTranspose1d.inference (layers/conv_layer.py:194-197).

    weight (Cin, Cout, K); pad length ceil(K/stride)-1 (:182); output cropped [stride:-stride].
    """
    P = pad_buffer.shape[-1]
    x = torch.cat((pad_buffer, x), -1)                       # :195
    new_pad = x[:, :, x.shape[-1] - P:]                      # :196
    y = F.conv_transpose1d(x, weight, bias, stride=stride, padding=0, output_padding=0)
    return y[:, :, stride:-stride], new_pad                  # :197


def causal_convtr1d_forward(x, weight, bias, stride):
    """CausalConvTranspose1d.forward: replication left-pad (layers/conv_layer.py:189-192)."""
    P = math.ceil(weight.shape[-1] / stride) - 1
    x = F.pad(x, (P, 0), mode="replicate")
    y = F.conv_transpose1d(x, weight, bias, stride=stride, padding=0, output_padding=0)
    return y[:, :, stride:-stride]


def effective_weight(sd, spec):
    """Weight the reference's conv actually multiplies by.

    Weight-normed modules recompute ``torch._weight_norm(v, g, dim=0)`` in a forward pre-hook
    (torch.nn.utils.weight_norm; applied at models/vocoder/HiFiGAN.py:182-191 and
    models/autoencoder/AudioDec.py:150-162).
    """
    if spec.wn:
        return torch._weight_norm(sd[spec.wkey("weight_v")], sd[spec.wkey("weight_g")], 0)
    return sd[spec.wkey("weight")]


# ----------------------------------------------------------------------------------------------
# layers/vq_module.py
# ----------------------------------------------------------------------------------------------
def vq_forward_index(inp, embed):
    """VectorQuantize.forward_index (layers/vq_module.py:90-104).  inp (..., 64); embed (64, 1024)."""
    flatten = inp.reshape(-1, embed.shape[0])
    dist = (
        flatten.pow(2).sum(1, keepdim=True)
        - 2 * flatten @ embed
        + embed.pow(2).sum(0, keepdim=True)
    )                                                         # :93-97
    _, embed_ind = (-dist).max(1)                             # :98
    embed_ind = embed_ind.view(*inp.shape[:-1])               # :100
    quantize = F.embedding(embed_ind, embed.transpose(0, 1))  # :101
    quantize = inp + (quantize - inp)                         # :102 (detach is a no-op for values)
    return quantize, embed_ind, dist


def rvq_forward_index(x, embeds, flatten_idx=True, return_margin=False):
    """ResidualVQ.forward_index (layers/vq_module.py:136-149) for ONE stream: x (1, T, 64).

    Returns (quantized_out, indices (n_q, T) int64[, top-2 distance margin (n_q, T)]).
    """
    codebook_size = embeds[0].shape[1]
    quantized_out = 0.0
    residual = x
    all_indices, margins = [], []
    for i, embed in enumerate(embeds):
        quantized, indices, dist = vq_forward_index(residual, embed)
        residual = residual - quantized                       # :143
        quantized_out = quantized_out + quantized             # :144
        if return_margin:
            top2 = torch.topk(-dist, 2, dim=1).values
            margins.append((top2[:, 0] - top2[:, 1]).reshape(indices.shape))
        if flatten_idx:
            indices = indices + codebook_size * i             # :145-146
        all_indices.append(indices)
    all_indices = torch.stack(all_indices).squeeze(1)         # :148-149
    if return_margin:
        return quantized_out, all_indices, torch.stack(margins).squeeze(1)
    return quantized_out, all_indices


def rvq_codebook(embeds):
    """ResidualVQ.initial (layers/vq_module.py:151-157): stacked (n_q*1024, 64) table."""
    cb = torch.stack([e.transpose(0, 1) for e in embeds])
    return cb.reshape(-1, cb.size(-1))


def rvq_lookup(indices, codebook):
    """ResidualVQ.lookup (layers/vq_module.py:159-161): (n_q, T) -> (1, T, 64)."""
    quantized_out = F.embedding(indices, codebook)
    return torch.sum(quantized_out, dim=0, keepdim=True)


# ----------------------------------------------------------------------------------------------
# shared plumbing: a bag of per-layer pad buffers
# ----------------------------------------------------------------------------------------------
class _Streaming:
    def __init__(self, sd, specs, batch=1):
        self.sd = {k: (v.detach().clone().float() if torch.is_floating_point(v) else v.clone())
                   for k, v in sd.items()}
        self.specs = {s.name: s for s in specs}
        self.batch = batch
        self.w = {s.name: effective_weight(self.sd, s) for s in specs}
        self.b = {s.name: (self.sd[s.wkey("bias")] if s.bias else None) for s in specs}
        self.pad = {}
        self.reset_buffer()

    def reset_buffer(self):
        """reset_buffer (AudioDec.py:250-256 / HiFiGAN.py:298-305): zero every pad_buffer."""
        for s in self.specs.values():
            if s.kind != "conv1x1":
                self.pad[s.name] = torch.zeros(self.batch, s.cin, s.pad)

    def load_pad_buffers(self):
        """Adopt the checkpoint's pad_buffer entries (what load_state_dict would do)."""
        for s in self.specs.values():
            k = f"{s.name}.pad_buffer"
            if s.kind != "conv1x1" and k in self.sd:
                self.pad[s.name] = self.sd[k].expand(self.batch, -1, -1).clone()

    # streaming / non-streaming conv by layer name
    def conv(self, name, x, streaming=True):
        s = self.specs[name]
        if s.kind == "conv1x1":
            return F.conv1d(x, self.w[name], self.b[name])
        if s.kind == "conv":
            if streaming:
                y, self.pad[name] = causal_conv1d_inference(
                    x, self.pad[name], self.w[name], self.b[name], s.stride, s.dilation, s.groups)
                return y
            return causal_conv1d_forward(x, self.w[name], self.b[name], s.stride, s.dilation, s.groups)
        if streaming:
            y, self.pad[name] = causal_convtr1d_inference(x, self.pad[name], self.w[name], self.b[name], s.stride)
            return y
        return causal_convtr1d_forward(x, self.w[name], self.b[name], s.stride)


# ----------------------------------------------------------------------------------------------
# models/autoencoder/*
# ----------------------------------------------------------------------------------------------
class AutoEncoderOracle(_Streaming):
    """models/autoencoder/AudioDec.py StreamGenerator (:166-256) with codec='audiodec' or
    'activate_audiodec'; B independent streams share the weights."""

    def __init__(self, sd, generator_params, batch=1):
        from audiodec_amd import arch
        self.p = generator_params
        specs = arch.autoencoder_encoder_convs(generator_params) + arch.autoencoder_decoder_convs(generator_params)
        super().__init__(sd, specs, batch)
        self.n_blocks = len(generator_params.get("enc_strides", (3, 4, 5, 5)))
        self.n_q = generator_params.get("codebook_num", 8)
        self.embeds = [self.sd[f"quantizer.codebook.layers.{i}.embed"] for i in range(self.n_q)]
        self.codebook = None
        self.activate = generator_params.get("codec", "audiodec") == "activate_audiodec"
        name = generator_params.get("nonlinear_activation", "ELU")
        self.act = getattr(torch.nn, name)(**generator_params.get("nonlinear_activation_params", {}))
        self.in_ch = generator_params.get("input_channels", 1)

    # residual_unit.py:43-46 (forward) / :78-81 (inference)
    def _res_unit(self, pre, x, streaming):
        y = self.conv(f"{pre}.conv1", self.act(x), streaming)
        y = self.conv(f"{pre}.conv2", self.act(y))
        return x + y

    # encoder.py:137-142 (encode) / :131-135 (forward), EncoderBlock :70-81
    def _encoder(self, x, streaming):
        x = self.conv("encoder.conv", x, streaming)
        for i in range(self.n_blocks):
            pre = f"encoder.conv_blocks.{i}"
            for j in range(3):
                x = self._res_unit(f"{pre}.res_units.{j}", x, streaming)
            x = self.conv(f"{pre}.conv", x, streaming)
        if self.activate:                                     # ActivateEncoder, encoder.py:171-175
            x = self.act(x)
        return x

    def encode(self, x, streaming=True):
        """StreamGenerator.encode (AudioDec.py:228-234)."""
        (batch, channel, length) = x.size()
        if channel != self.in_ch:
            x = x.reshape(-1, self.in_ch, length)
        x = self._encoder(x, streaming)
        return self.conv("projector.project", x, streaming)   # projector.py:52-54

    def quantize(self, z, return_margin=False):
        """StreamGenerator.quantize (AudioDec.py:237-239) -> Quantizer.encode (quantizer.py:42-44).

        The reference only handles B == 1 (vq_module.py:148-149); B streams = B reference calls.
        Returns (n_q, T) for B == 1, else (n_q, B, T).
        """
        outs, margins = [], []
        for b in range(z.shape[0]):
            r = rvq_forward_index(z[b:b + 1].transpose(2, 1), self.embeds, True, return_margin)
            outs.append(r[1])
            if return_margin:
                margins.append(r[2])
        idx = outs[0] if len(outs) == 1 else torch.stack(outs, 1)
        if return_margin:
            return idx, (margins[0] if len(margins) == 1 else torch.stack(margins, 1))
        return idx

    def initial(self):
        self.codebook = rvq_codebook(self.embeds)             # quantizer.py:29-30

    def lookup(self, idx):
        """StreamGenerator.lookup (AudioDec.py:242-243): (n_q,T)->(1,T,64); (n_q,B,T)->(B,T,64)."""
        if self.codebook is None:
            self.initial()
        if idx.dim() == 2:
            return rvq_lookup(idx, self.codebook)
        return torch.cat([rvq_lookup(idx[:, b], self.codebook) for b in range(idx.shape[1])], 0)

    # decoder.py:142-148 (decode) / :136-140 (forward); DecoderBlock :70-81; ActivateDecoder :203-214
    def _decoder(self, z, streaming):
        x = self.conv("decoder.conv1", z, streaming)
        for i in range(self.n_blocks):
            pre = f"decoder.conv_blocks.{i}.1" if self.activate else f"decoder.conv_blocks.{i}"
            if self.activate:
                x = self.act(x)
            x = self.conv(f"{pre}.conv", x, streaming)
            for j in range(3):
                x = self._res_unit(f"{pre}.res_units.{j}", x, streaming)
        if self.activate:
            x = self.act(x)
        x = self.conv("decoder.conv2", x, streaming)
        if self.activate:
            x = torch.tanh(x)
        return x

    def decode(self, zq, streaming=True):
        """StreamGenerator.decode (AudioDec.py:246-247): zq (B, T, 64) -> (B, out, T*hop)."""
        return self._decoder(zq.transpose(2, 1), streaming)

    def initial_encoder(self, receptive_length=8192):
        """AudioDec.py:216-221: warm-up with silence; fills every encoder-side pad buffer."""
        self.initial()
        z = self.encode(torch.zeros(self.batch, self.in_ch, receptive_length))
        idx = self.quantize(z[:1])
        return self.lookup(idx)

    def initial_decoder(self, zq):
        self.decode(zq.expand(self.batch, -1, -1) if zq.shape[0] != self.batch else zq)   # :224-225

    def forward_quantized(self, x):
        """Generator.forward pieces (AudioDec.py:112-120) with inference-time quantiser."""
        z = self.encode(x, streaming=False)
        idx = self.quantize(z)
        zq = self.lookup(idx)
        return z, idx, zq, self.decode(zq, streaming=False)

    def analyze(self, x):
        """What codecTest.py:78-88 / codecStatistic.py:92-97 run: ``encoder`` -> ``projector`` -> ``quantizer``
        forwards (eval mode); x (B, 1, L) -> zq (B, 64, T').  Quantizer.forward (quantizer.py:32-35) ->
        ResidualVQ.forward (vq_module.py:119-134) does the same value arithmetic as forward_index."""
        z = self.encode(x, streaming=False)
        zq = torch.cat([rvq_forward_index(z[b:b + 1].transpose(2, 1), self.embeds)[0] for b in range(z.shape[0])], 0)
        return zq.transpose(2, 1)

    def synthesize(self, zq):
        """codecTest.py:90-95 for a symAudioDec decoder: ``decoder.decoder(zq)`` (Decoder.forward)."""
        return self._decoder(zq, streaming=False)


# ----------------------------------------------------------------------------------------------
# models/vocoder/HiFiGAN.py (+ modules/residual_block.py, modules/multi_fusion.py)
# ----------------------------------------------------------------------------------------------
class HiFiGANOracle(_Streaming):
    """models/vocoder/HiFiGAN.py StreamGenerator (:222-305)."""

    def __init__(self, sd, generator_params, batch=1):
        from audiodec_amd import arch
        self.p = generator_params
        super().__init__(sd, arch.hifigan_convs(generator_params), batch)
        self.multigroup = arch.hifigan_is_multigroup(generator_params)
        self.groups = generator_params.get("groups", 1)
        self.n_up = len(generator_params["upsample_scales"])
        self.n_rb = len(generator_params["resblock_kernel_sizes"])
        self.n_dil = [len(d) for d in generator_params["resblock_dilations"]]
        self.addl = generator_params.get("use_additional_convs", True)
        act = generator_params.get("nonlinear_activation", "LeakyReLU")
        self.act = getattr(torch.nn, act)(**generator_params.get("nonlinear_activation_params", {"negative_slope": 0.1}))
        self.act_out1 = torch.nn.LeakyReLU()                  # HiFiGAN.py:116 (default slope 0.01)
        self.norm = "mean" in self.sd                         # HiFiGAN.py:126-131
        self.mean = self.sd.get("mean")
        self.scale = self.sd.get("scale")

    def _resblock(self, pre, x, n_layer, streaming):
        """HiFiGANResidualBlock.inference (residual_block.py:99-105) / forward (:84-97)."""
        for idx in range(n_layer):
            xt = self.conv(f"{pre}.convs1.{idx}", self.act(x), streaming)
            if self.addl:
                xt = self.conv(f"{pre}.convs2.{idx}", self.act(xt), streaming)
            x = xt + x
        return x

    def _fusion(self, i, c, streaming):
        if self.multigroup:                                   # multi_fusion.py:133-141
            x = c.repeat(1, self.groups, 1)
            x = self._resblock(f"blocks.{i}", x, self.n_dil[0], streaming)
            return self.conv(f"blocks.{i}.conv_out", x)
        cs = 0.0                                              # multi_fusion.py:73-79
        for b in range(self.n_rb):
            cs += self._resblock(f"blocks.{i}.blocks.{b}", c, self.n_dil[b], streaming)
        return cs / self.n_rb

    def decode(self, c, streaming=True):
        """StreamGenerator.decode (HiFiGAN.py:268-296): c (B, T, 64) -> (B, 1, T*hop)."""
        if self.norm:
            c = (c - self.mean) / self.scale                  # :276-279
        c = self.conv("input_conv", c.transpose(2, 1), streaming)          # :282-284
        for i in range(self.n_up):                            # :287-291
            c = self.conv(f"upsamples.{i}", self.act(c), streaming)
            c = self._fusion(i, c, streaming)
        c = self.conv("output_conv", self.act_out1(c), streaming)          # :294-296
        return torch.tanh(c)

    def initial_decoder(self, c):
        self.decode(c.expand(self.batch, -1, -1) if c.shape[0] != self.batch else c)      # :264-265

    def synthesize(self, zq):
        """codecTest.py:90-95 for a vocoder decoder: Generator.forward (HiFiGAN.py:141-161); zq (B, 64, T')."""
        return self.decode(zq.transpose(2, 1), streaming=False)


def build_decoder_oracle(sd, model_type, generator_params, batch=1):
    """utils/audiodec.py:44-56 dispatch."""
    if model_type in ("symAudioDec", "symAudioDecUniv"):
        return AutoEncoderOracle(sd, generator_params, batch)
    if model_type in ("HiFiGAN", "UnivNet"):
        return HiFiGANOracle(sd, generator_params, batch)
    raise NotImplementedError(f"Decoder {model_type} is not supported!")
