"""Layer-level mirrors of /root/reference/layers/conv_layer.py and layers/vq_module.py.

``CausalConv1d`` / ``CausalConvTranspose1d`` expose the reference's constructor arguments and the
streaming ``inference`` / ``reset_buffer`` methods (conv_layer.py:118-159, 162-200) on top of
``adk_causal_conv`` with a private state ring; ``ResidualVQ`` exposes ``forward_index`` / ``initial``
/ ``lookup`` (vq_module.py:136-161) on top of ``adk_rvq_encode`` / ``adk_rvq_lookup``.  The model
programs (program.py) do not go through these objects -- they exist so single layers can be used and
tested against the reference's layer classes one to one.

Tensors keep the reference's (B, C, T) convention at this boundary; inside, rows are channel-last.
``inference`` needs ``L % stride == 0`` (the reference tolerates ragged chunks but loses stride phase,
SURVEY.md appendix C).
"""
import ctypes as C
import math

import torch

from . import native
from .native import ConvDesc, RingView
from .program import pack_conv, pack_convtr, pack_mfma, pack_split16, mfma_eligible, split16_eligible

_ACTS = {None: native.ACT_NONE, "ELU": native.ACT_ELU, "LeakyReLU": native.ACT_LEAKY, "Tanh": native.ACT_TANH}


def _view(t, rows, channels, cursor, ch_off=0):
    v = RingView()
    v.base, v.rows, v.channels, v.cursor, v.ch_off = (t.data_ptr() if t is not None else None), rows, channels, cursor, ch_off
    return v


class _CausalBase:
    def __init__(self, in_channels, hist, device, batch, max_len):
        self.dev = native.require_gpu(device)
        self.batch, self.hist, self.max_len = batch, hist, max_len
        self.in_channels = in_channels
        self.rows = hist + max_len
        self.ring = torch.zeros(batch, self.rows, in_channels, device=self.dev)
        self.cursor = 0
        self.impl = native.IMPL_AUTO
        self.act_in, self.slope, self.act_out = native.ACT_NONE, 0.0, native.ACT_NONE

    def set_activation(self, act_in=None, slope=0.0, act_out=None):
        """Fuse the activation the reference applies before / after this conv."""
        self.act_in, self.slope, self.act_out = _ACTS[act_in], float(slope), _ACTS[act_out]
        return self

    def reset_buffer(self):
        self.ring.zero_()          # conv_layer.py:158-159 / :199-200
        self.cursor = 0

    @property
    def pad_buffer(self):
        """The reference's pad_buffer view of the state: (B, Cin, P), oldest first."""
        idx = (self.cursor - self.hist + torch.arange(self.hist, device=self.dev)) % self.rows
        return self.ring[:, idx, :].transpose(1, 2)

    def _push(self, x):
        B, Cin, L = x.shape
        if B != self.batch or Cin != self.in_channels:
            raise ValueError(f"expected ({self.batch}, {self.in_channels}, L), got {tuple(x.shape)}")
        if L > self.max_len:
            raise ValueError(f"chunk of {L} steps exceeds max_len={self.max_len}")
        src = x.to(self.dev, torch.float32).transpose(1, 2).contiguous()
        native.check(native.lib().adk_ring_write(C.c_void_p(src.data_ptr()), _view(self.ring, self.rows, Cin, self.cursor),
                                                 None, None, B, L, native.current_stream(self.dev)), "adk_ring_write")
        return L

    def _run(self, d, t_out, out_rows, out_ch, residual=None, time_iters=0):
        out = torch.empty(self.batch, out_rows, out_ch, device=self.dev)
        res_view = _view(None, 0, 0, 0)
        if residual is not None:
            # fused epilogue add (x + conv(...): residual_unit.py:78-81, residual_block.py:103): residual (B, Cout, T)
            if tuple(residual.shape) != (self.batch, out_ch, t_out) or out_rows != t_out:
                raise ValueError(f"residual must be ({self.batch}, {out_ch}, {t_out})")
            res = residual.to(self.dev, torch.float32).transpose(1, 2).contiguous()
            res_view = _view(res, t_out, out_ch, 0)
        d.act_in, d.act_in_slope, d.act_out = self.act_in, self.slope, self.act_out
        d.w = self.w_packed.data_ptr()
        d.w_frag = self.w_frag.data_ptr() if self.w_frag is not None else None
        if self.impl in (native.IMPL_SPLIT16, native.IMPL_SPLIT16_ROWS, native.IMPL_SPLIT16_SK, native.IMPL_SPLIT16_UP):
            if getattr(self, "w_split", None) is None:
                raise ValueError("this layer shape has no split-f16 kernel")
            d.w_frag = self.w_split.data_ptr()
        d.bias = self.b_packed.data_ptr() if self.b_packed is not None else None
        in_view, out_view = _view(self.ring, self.rows, self.in_channels, self.cursor), _view(out, out_rows, out_ch, 0)
        buf = C.create_string_buffer(64)
        native.check(native.lib().adk_causal_conv_describe(C.byref(d), in_view, out_view, res_view, self.batch, t_out, self.impl, buf, 64),
                     "adk_causal_conv_describe")
        self.last_kernel = buf.value.decode()          # which kernel this call runs (tests, profiles)
        if time_iters:
            us = C.c_float(0.0)
            native.check(native.lib().adk_causal_conv_time(C.byref(d), in_view, out_view, res_view, self.batch, t_out, self.impl, int(time_iters),
                                                           native.current_stream(self.dev), C.byref(us)), "adk_causal_conv_time")
            return float(us.value)
        native.check(native.lib().adk_causal_conv(
            C.byref(d), in_view, out_view, res_view, self.batch, t_out, self.impl, native.current_stream(self.dev)), "adk_causal_conv")
        return out


class CausalConv1d(_CausalBase):
    """1D causal convolution w/ 1-side padding (layers/conv_layer.py:118-159)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride=1, dilation=1, groups=1, bias=True,
                 device="cuda:0", batch=1, max_len=4096):
        super().__init__(in_channels, (kernel_size - 1) * dilation, device, batch, max_len)
        self.out_channels, self.kernel_size = out_channels, kernel_size
        self.stride, self.dilation, self.groups = stride, dilation, groups
        self.pad_length = self.hist
        self.weight = torch.zeros(out_channels, in_channels // groups, kernel_size)
        self.bias = torch.zeros(out_channels) if bias else None
        self.load(self.weight, self.bias)

    def load(self, weight, bias=None):
        assert tuple(weight.shape) == (self.out_channels, self.in_channels // self.groups, self.kernel_size)
        self.weight, self.bias = weight.detach().float().cpu(), (bias.detach().float().cpu() if bias is not None else None)
        rows = pack_conv(self.weight)
        self.w_packed = rows.to(self.dev)
        ok = mfma_eligible(self.in_channels // self.groups, self.out_channels // self.groups, self.groups)
        self.w_frag = pack_mfma(rows, self.groups).to(self.dev) if ok else None
        ok16 = split16_eligible(self.in_channels // self.groups, self.out_channels // self.groups, self.groups)
        self.w_split = pack_split16(rows, self.groups).to(self.dev) if ok16 else None
        self.b_packed = self.bias.to(self.dev) if self.bias is not None else None
        return self

    def inference(self, x, residual=None):
        """x (B, Cin, L) -> (B, Cout, L/stride); `residual` (B, Cout, L/stride), if given, is added in the kernel's epilogue."""
        L = self._push(x)
        if L % self.stride:
            raise ValueError(f"chunk length {L} is not a multiple of the stride {self.stride}")
        t_out = L // self.stride
        d = ConvDesc()
        d.cin_g, d.cout_g, d.groups = self.in_channels // self.groups, self.out_channels // self.groups, self.groups
        d.taps, d.stride, d.dilation, d.hist = self.kernel_size, self.stride, self.dilation, self.hist
        d.up, d.cout_real = 1, self.out_channels
        d.in_group_stride, d.res_group_stride = d.cin_g, d.cout_g
        out = self._run(d, t_out, t_out, self.out_channels, residual)
        self.cursor = (self.cursor + L) % self.rows
        return out.transpose(1, 2)


class CausalConvTranspose1d(_CausalBase):
    """1D causal transposed convolution, kernel = 2*stride (layers/conv_layer.py:162-200)."""

    def __init__(self, in_channels, out_channels, kernel_size, stride, bias=True, device="cuda:0", batch=1, max_len=4096):
        if kernel_size != 2 * stride:
            raise NotImplementedError("the streaming path only uses kernel_size == 2*stride (HiFiGAN.py:95, decoder.py:53)")
        super().__init__(in_channels, math.ceil(kernel_size / stride) - 1, device, batch, max_len)
        self.out_channels, self.kernel_size, self.stride = out_channels, kernel_size, stride
        self.pad_length = self.hist
        self.load(torch.zeros(in_channels, out_channels, kernel_size), torch.zeros(out_channels) if bias else None)

    def load(self, weight, bias=None):
        assert tuple(weight.shape) == (self.in_channels, self.out_channels, self.kernel_size)
        self.weight, self.bias = weight.detach().float().cpu(), (bias.detach().float().cpu() if bias is not None else None)
        rows = pack_convtr(self.weight, self.stride)
        self.w_packed = rows.to(self.dev)
        ok = mfma_eligible(self.in_channels, self.stride * self.out_channels, 1)
        self.w_frag = pack_mfma(rows, 1).to(self.dev) if ok else None
        self.w_split = pack_split16(rows, 1).to(self.dev) if split16_eligible(self.in_channels, self.stride * self.out_channels, 1) else None
        self.b_packed = self.bias.repeat(self.stride).to(self.dev) if self.bias is not None else None
        return self

    def _desc(self):
        d = ConvDesc()
        d.cin_g, d.cout_g, d.groups = self.in_channels, self.stride * self.out_channels, 1
        d.taps, d.stride, d.dilation, d.hist = 2, 1, 1, 1
        d.up, d.cout_real = self.stride, self.out_channels
        d.in_group_stride, d.res_group_stride = d.cin_g, d.cout_g
        return d

    def inference(self, x):
        L = self._push(x)
        out = self._run(self._desc(), L, L * self.stride, self.out_channels)
        self.cursor = (self.cursor + L) % self.rows
        return out.transpose(1, 2)

    def time_kernel(self, L, iters=300):
        """Average microseconds of the conv launch of inference() alone, on the rows already in the ring (no new input, no state
        change): `iters` back-to-back launches timed by HIP events inside the library (adk_causal_conv_time) -- bench.py's
        roofline of the last up-sampling stage."""
        return self._run(self._desc(), L, L * self.stride, self.out_channels, time_iters=iters)


class ResidualVQ:
    """Residual VQ inference (layers/vq_module.py:107-161): forward_index / initial / lookup."""

    def __init__(self, embeds, device="cuda:0"):
        """embeds: list of the reference's `embed` buffers, each (dim, codebook_size)."""
        self.dev = native.require_gpu(device)
        self.n_q = len(embeds)
        self.dim, self.codebook_size = embeds[0].shape
        emb = [e.detach().float().cpu() for e in embeds]
        self.embed = torch.stack(emb).contiguous().to(self.dev)
        self.enorm = torch.stack([e.pow(2).sum(0, keepdim=True)[0] for e in emb]).contiguous().to(self.dev)
        self._emb_cpu = emb
        self.codebook = None

    def forward_index(self, x, flatten_idx=False):
        """x (B, T, dim) -> (quantized_out (B, T, dim), indices (n_q, B, T) squeezed at dim 1)."""
        B, T, D = x.shape
        xt = x.to(self.dev, torch.float32).contiguous()
        idx = torch.empty(self.n_q, B * T, dtype=torch.int64, device=self.dev)
        zq = torch.empty(B, T, D, device=self.dev)
        native.check(native.lib().adk_rvq_encode(
            C.c_void_p(xt.data_ptr()), C.c_void_p(self.embed.data_ptr()), C.c_void_p(self.enorm.data_ptr()),
            C.c_void_p(idx.data_ptr()), C.c_void_p(zq.data_ptr()), B * T, self.n_q, self.dim, self.codebook_size,
            native.current_stream(self.dev)), "adk_rvq_encode")
        idx = idx.reshape(self.n_q, B, T)
        if not flatten_idx:
            idx = idx - (torch.arange(self.n_q, device=self.dev) * self.codebook_size).view(-1, 1, 1)
        return zq, idx.squeeze(1)

    def initial(self):
        cb = torch.stack([e.transpose(0, 1) for e in self._emb_cpu])
        self.codebook = cb.reshape(-1, cb.size(-1)).contiguous().to(self.dev)

    def lookup(self, indices):
        if self.codebook is None:
            raise AttributeError("call initial() first (layers/vq_module.py:151-157)")
        idx = indices.to(self.dev, torch.int64)
        if idx.dim() == 2:
            idx = idx.unsqueeze(1)
        n_q, B, T = idx.shape
        idx = idx.contiguous()
        zq = torch.empty(B, T, self.dim, device=self.dev)
        native.check(native.lib().adk_rvq_lookup(
            C.c_void_p(idx.data_ptr()), C.c_void_p(self.codebook.data_ptr()), C.c_void_p(zq.data_ptr()), B * T, n_q,
            self.dim, self.codebook.shape[0], native.current_stream(self.dev)), "adk_rvq_lookup")
        return zq
