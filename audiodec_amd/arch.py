"""Static description of the convolutions on the AudioDec streaming path.

Given a ``generator_params`` dict (the block the reference passes to
``StreamGenerator(**config['generator_params'])``, /root/reference/utils/audiodec.py:40,54) this
module enumerates every convolution with the state-dict key prefix the reference gives it, so the
checkpoint loader, the synthetic-checkpoint writer and the program builder agree on names/shapes.

Reference structure restated here:
  autoencoder  models/autoencoder/AudioDec.py:61-104, modules/encoder.py:25-134,
               modules/decoder.py:25-138, modules/residual_unit.py:20-76, modules/projector.py:20-47
  vocoder      models/vocoder/HiFiGAN.py:71-117, modules/multi_fusion.py:23-112,
               modules/residual_block.py:23-79
  layers       layers/conv_layer.py:118-200 (CausalConv1d / CausalConvTranspose1d), :28-32 (1x1)
"""
from dataclasses import dataclass
from typing import List, Optional


@dataclass
class ConvSpec:
    name: str            # module path; weights live at f"{name}.{sub}weight" (see wkey())
    kind: str            # 'conv' (CausalConv1d) | 'convT' (CausalConvTranspose1d) | 'conv1x1'
    cin: int
    cout: int
    k: int
    stride: int = 1
    dilation: int = 1
    groups: int = 1
    bias: bool = True
    wn: bool = False     # weight-normed (weight_g / weight_v)

    @property
    def pad(self) -> int:
        """Streaming history length (conv_layer.py:141, :182)."""
        if self.kind == "conv":
            return (self.k - 1) * self.dilation
        if self.kind == "convT":
            return -(-self.k // self.stride) - 1
        return 0

    @property
    def sub(self) -> str:
        return {"conv": "conv.", "convT": "deconv.", "conv1x1": ""}[self.kind]

    def wkey(self, what="weight") -> str:
        return f"{self.name}.{self.sub}{what}"

    @property
    def wshape(self):
        if self.kind == "convT":
            return (self.cin, self.cout // self.groups, self.k)
        return (self.cout, self.cin // self.groups, self.k)

    @property
    def macs_per_out(self) -> int:
        return (self.cin // self.groups) * self.k * self.cout


def _res_units(prefix, c, wn, dilations=(1, 3, 9)) -> List[ConvSpec]:
    out = []
    for j, d in enumerate(dilations):
        out.append(ConvSpec(f"{prefix}.res_units.{j}.conv1", "conv", c, c, 7, 1, d, 1, False, wn))
        out.append(ConvSpec(f"{prefix}.res_units.{j}.conv2", "conv1x1", c, c, 1, 1, 1, 1, False, wn))
    return out


def autoencoder_encoder_convs(p) -> List[ConvSpec]:
    wn = bool(p.get("use_weight_norm", False))
    ch = p.get("encode_channels", 32)
    ratios = p.get("enc_ratios", (2, 4, 8, 16))
    strides = p.get("enc_strides", (3, 4, 5, 5))
    bias = p.get("bias", True)
    out = [ConvSpec("encoder.conv", "conv", p.get("input_channels", 1), ch, 7, 1, 1, 1, False, wn)]
    cin = ch
    for i, s in enumerate(strides):
        cout = ch * ratios[i]
        pre = f"encoder.conv_blocks.{i}"
        out += _res_units(pre, cin, wn)
        out.append(ConvSpec(f"{pre}.conv", "conv", cin, cout, 2 * s, s, 1, 1, bias, wn))
        cin = cout
    out.append(ConvSpec("projector.project", "conv", cin, p.get("code_dim", 64), 3, 1, 1, 1, False, wn))
    return out


def autoencoder_decoder_convs(p) -> List[ConvSpec]:
    wn = bool(p.get("use_weight_norm", False))
    activate = p.get("codec", "audiodec") == "activate_audiodec"
    ch = p.get("decode_channels", 32)
    ratios = p.get("dec_ratios", (16, 8, 4, 2))
    strides = p.get("dec_strides", (5, 5, 4, 3))
    bias = p.get("bias", True)
    out = [ConvSpec("decoder.conv1", "conv", p.get("code_dim", 64), ch * ratios[0], 7, 1, 1, 1, False, wn)]
    cout = ch
    for i, s in enumerate(strides):
        cin = ch * ratios[i]
        cout = ch * ratios[i + 1] if i < len(ratios) - 1 else ch
        # ActivateDecoder wraps each block in Sequential(act, DecoderBlock) (decoder.py:186-201)
        pre = f"decoder.conv_blocks.{i}.1" if activate else f"decoder.conv_blocks.{i}"
        out.append(ConvSpec(f"{pre}.conv", "convT", cin, cout, 2 * s, s, 1, 1, bias, wn))
        out += _res_units(pre, cout, wn)
    out.append(ConvSpec("decoder.conv2", "conv", cout, p.get("output_channels", 1), 7, 1, 1, 1, False, wn))
    return out


def hifigan_is_multigroup(p) -> bool:
    ks, ds = p["resblock_kernel_sizes"], p["resblock_dilations"]
    return len(ks) == len(ds) == 1 and p.get("groups", 1) > 1      # HiFiGAN.py:66-69


def hifigan_convs(p) -> List[ConvSpec]:
    wn = bool(p.get("use_weight_norm", True))
    ch = p.get("channels", 512)
    bias = p.get("bias", True)
    groups = p.get("groups", 1)
    ks, ds = p["resblock_kernel_sizes"], p["resblock_dilations"]
    addl = p.get("use_additional_convs", True)
    out = [ConvSpec("input_conv", "conv", p["in_channels"], ch, p.get("kernel_size", 7), 1, 1, 1, True, wn)]
    c = ch
    for i, (s, uk) in enumerate(zip(p["upsample_scales"], p["upsample_kernel_sizes"])):
        assert uk == 2 * s
        cin, c = ch // (2 ** i), ch // (2 ** (i + 1))
        out.append(ConvSpec(f"upsamples.{i}", "convT", cin, c, uk, s, 1, 1, True, wn))
        if hifigan_is_multigroup(p):
            cg = c * groups
            for j, d in enumerate(ds[0]):
                out.append(ConvSpec(f"blocks.{i}.convs1.{j}", "conv", cg, cg, ks[0], 1, d, groups, bias, wn))
                if addl:
                    out.append(ConvSpec(f"blocks.{i}.convs2.{j}", "conv", cg, cg, ks[0], 1, 1, groups, bias, wn))
            out.append(ConvSpec(f"blocks.{i}.conv_out", "conv1x1", cg, c, 1, 1, 1, 1, False, wn))
        else:
            for b, (k, dil) in enumerate(zip(ks, ds)):
                for j, d in enumerate(dil):
                    out.append(ConvSpec(f"blocks.{i}.blocks.{b}.convs1.{j}", "conv", c, c, k, 1, d, groups, bias, wn))
                    if addl:
                        out.append(ConvSpec(f"blocks.{i}.blocks.{b}.convs2.{j}", "conv", c, c, k, 1, 1, groups, bias, wn))
    out.append(ConvSpec("output_conv", "conv", c, p.get("out_channels", 1), p.get("kernel_size", 7), 1, 1, 1, True, wn))
    return out


def by_name(specs: List[ConvSpec]):
    return {s.name: s for s in specs}


def hop_length(p) -> int:
    import math
    if "enc_strides" in p:
        return math.prod(p["enc_strides"])
    return math.prod(p["upsample_scales"])
