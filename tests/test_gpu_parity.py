"""GPU parity: the HIP path (through the C ABI) against the reference's committed outputs
(tests/golden/*.npz) and against the CPU oracle on seeded inputs.

Tolerances (north-star): decoded waveform <= 1e-4 max-abs; RVQ indices bit-exact -- a mismatch is
reported with the reference's own top-2 distance margin so it can be told apart from a bug.
"""
import os

import numpy as np
import pytest
import torch

from audiodec_amd import configs, synth
from oracle import audiodec_oracle as O
import op_cases as C
from test_oracle_golden import golden_audio, build_oracle, explain_flips, golden_chunks

pytestmark = pytest.mark.gpu

WAVE_TOL = 1e-4
DEV = "cuda:0"


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"{name}.npz"), allow_pickle=False)


def load_audiodec(ckpt_root, model, seed, num_streams, max_frames, split16=False, guard=None):
    from audiodec_amd.audiodec import AudioDec
    from audiodec_amd.configs import checkpoint_paths as assign_model      # also knows the EXTRA_ALIASES test models
    synth.write_model(ckpt_root, model, seed)
    cwd = os.getcwd()
    os.chdir(ckpt_root)          # the reference's paths are cwd-relative ('exp/...', 'stats/...')
    old = os.environ.get("ADK_SPLIT16")
    os.environ["ADK_SPLIT16"] = "1" if split16 else "0"       # read by the generators when they are constructed
    try:
        sr, enc_ckpt, dec_ckpt = assign_model(model)
        ad = AudioDec(tx_device=DEV, rx_device=DEV, num_streams=num_streams, max_frames=max_frames, guard=guard)
        ad.load_transmitter(enc_ckpt)
        ad.load_receiver(enc_ckpt, dec_ckpt)
    finally:
        os.chdir(cwd)
        if old is None:
            del os.environ["ADK_SPLIT16"]
        else:
            os.environ["ADK_SPLIT16"] = old
    assert ad.tx_encoder.split16 == split16 and ad.decoder.split16 == split16
    return ad


# ------------------------------------------------------------------------------------------------
# layer level, against the reference's layer outputs (ops.npz) -- both kernels
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("impl", ["direct", "mfma"])
def test_causal_conv1d_matches_reference_layers(gpu, golden_dir, impl):
    from audiodec_amd import layers, native
    g = _load(golden_dir, "ops")
    ran = 0
    for n, (ci, co, k, s, d, gr, b, L1, L2) in enumerate(C.CONVS):
        if impl == "mfma" and ((ci // gr) % 32 or co % 4):
            continue
        x1, x2, w, bias = C.conv_inputs(n)
        m = layers.CausalConv1d(ci, co, k, s, d, gr, b, device=gpu, batch=1, max_len=256).load(w, bias)
        m.impl = native.IMPL_MFMA if impl == "mfma" else native.IMPL_DIRECT
        y1 = m.inference(x1).cpu().numpy()
        y2 = m.inference(x2).cpu().numpy()
        assert np.abs(y1 - g[f"conv{n}_y1"]).max() < 1e-5, (n, impl)
        assert np.abs(y2 - g[f"conv{n}_y2"]).max() < 1e-5, (n, impl)
        assert np.array_equal(m.pad_buffer.cpu().numpy(), g[f"conv{n}_pad"]), (n, impl)   # state is the raw input
        ran += 1
    assert ran >= 3


@pytest.mark.parametrize("impl", ["direct", "mfma", "split16_sk", "split16_rows", "split16_up"])
def test_causal_convtranspose1d_matches_reference_layers(gpu, golden_dir, impl):
    from audiodec_amd import layers, native
    g = _load(golden_dir, "ops")
    for n, (ci, co, s, L1, L2) in enumerate(C.CONVTS):
        if impl == "split16_rows" and (ci not in (32, 64) or (s * co) % 32):
            continue                                          # the rows-in-LDS kernel takes 32 / 64 input channels
        if impl == "split16_up" and (ci != 64 or (s * co) % 32 or s * co > 96):
            continue                                          # the up-sampling streamer: 64 input channels, <= 96 GEMM rows
        x1, x2, w, bias = C.convt_inputs(n)
        m = layers.CausalConvTranspose1d(ci, co, 2 * s, s, device=gpu, batch=1, max_len=64).load(w, bias)
        m.impl = {"mfma": native.IMPL_MFMA, "direct": native.IMPL_DIRECT, "split16_sk": native.IMPL_SPLIT16_SK,
                  "split16_rows": native.IMPL_SPLIT16_ROWS, "split16_up": native.IMPL_SPLIT16_UP}[impl]
        y1 = m.inference(x1).cpu().numpy()
        y2 = m.inference(x2).cpu().numpy()
        assert np.abs(y1 - g[f"convT{n}_y1"]).max() < 1e-5, (n, impl)
        assert np.abs(y2 - g[f"convT{n}_y2"]).max() < 1e-5, (n, impl)
        assert np.array_equal(m.pad_buffer.cpu().numpy(), g[f"convT{n}_pad"])


def test_residual_vq_matches_reference_layers(gpu, golden_dir):
    from audiodec_amd import layers
    g = _load(golden_dir, "ops")
    embeds, x = C.rvq_inputs()
    rvq = layers.ResidualVQ(embeds, device=gpu)
    q, idx = rvq.forward_index(x, flatten_idx=True)
    assert np.array_equal(idx.cpu().numpy(), g["rvq_idx"])             # bit-exact, incl. the engineered tie
    assert int(idx[0, 7]) == 123
    assert np.abs(q.cpu().numpy() - g["rvq_q"]).max() < 1e-5
    rvq.initial()
    zq = rvq.lookup(idx)
    assert np.abs(zq.cpu().numpy() - g["rvq_zq"]).max() < 1e-6
    # un-flattened indices (ResidualVQ.forward_index default)
    _, idx0 = rvq.forward_index(x)
    assert np.array_equal(idx0.cpu().numpy(), g["rvq_idx"] - 1024 * np.arange(4)[:, None])


# ------------------------------------------------------------------------------------------------
# fused activation / grouped / residual paths against the oracle on random tensors
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("cin,cout,k,s,d,gr,act,B,L", [
    (64, 64, 7, 1, 3, 1, "ELU", 3, 50), (96, 96, 11, 1, 5, 3, "LeakyReLU", 5, 40), (128, 256, 10, 5, 1, 1, None, 7, 25),
    (512, 64, 3, 1, 1, 1, "ELU", 33, 1), (32, 96, 3, 1, 1, 1, "LeakyReLU", 2, 301), (192, 192, 11, 1, 5, 3, "LeakyReLU", 2, 500),
    (32, 32, 7, 1, 9, 1, "ELU", 4, 333)])
def test_conv_kernels_agree_with_oracle(gpu, cin, cout, k, s, d, gr, act, B, L):
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(cin * 7 + k)
    w = torch.randn(cout, cin // gr, k, generator=g) / (cin // gr * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    fn = {None: lambda v: v, "ELU": torch.nn.ELU(), "LeakyReLU": torch.nn.LeakyReLU(0.1)}[act]
    pad = torch.zeros(B, cin, (k - 1) * d)
    mods = []
    impls = [native.IMPL_DIRECT, native.IMPL_MFMA]
    if s == 1 and cin // gr in (32, 64) and (cout // gr) % 32 == 0 and L >= 24:
        impls.append(native.IMPL_MFMA_ROWS)              # the rows-in-LDS kernel takes this shape
    if s == 1 and cin // gr in (32, 64) and (cout // gr) % 32 == 0 and k in (3, 7, 11):
        impls.append(native.IMPL_SPLIT16_ROWS)           # ... and its split-f16 variant
    impls.append(native.IMPL_SPLIT16_SK)                 # split-f16 stream-K takes every matrix-core shape
    for impl in impls:
        m = layers.CausalConv1d(cin, cout, k, s, d, gr, True, device=gpu, batch=B, max_len=L * s).load(w, bias)
        m.set_activation(act, 0.1)
        m.impl = impl
        mods.append(m)
    for step in range(3):                                 # ring wraps around during these steps
        x = torch.randn(B, cin, L * s, generator=g)
        ref, pad = O.causal_conv1d_inference(fn(x), fn(pad) if step == 0 else pad, w, bias, s, d, gr)
        for m in mods:
            y = m.inference(x).cpu()
            assert float((y - ref).abs().max()) < 2e-5, (m.impl, step)


@pytest.mark.parametrize("cin,cout,k,s,d,gr,act,B,L,what", [
    (768, 768, 11, 1, 5, 3, "LeakyReLU", 256, 5, "240 tiles x 44 chunks: two workgroups per tile, owners on the late blocks with 45 % of the tile"),
    (128, 128, 7, 1, 9, 1, "ELU", 256, 25, "200 tiles x 14 chunks: exact halves, 400 workgroups"),
    (256, 256, 7, 1, 9, 1, "ELU", 64, 5, "20 tiles x 28 chunks: five workgroups per tile"),
    (128, 192, 3, 1, 1, 1, None, 37, 5, "9 tiles x 6 chunks: three per tile, 27 workgroups (not a multiple of 8)"),
    (64, 128, 1, 1, 1, 1, None, 300, 100, "938 one-chunk tiles: several whole tiles per workgroup"),
    (384, 128, 1, 1, 1, 1, None, 256, 25, "200 tiles x 6 chunks: one whole tile per workgroup"),
    (384, 384, 11, 1, 5, 3, "LeakyReLU", 256, 25, "300 tiles x 22 chunks: balanced split, ranges cut anywhere"),
    (512, 1280, 2, 1, 1, 1, "LeakyReLU", 256, 1, "80 tiles x 16 chunks: three per tile"),
    (64, 64, 3, 1, 1, 1, "ELU", 1, 3, "one tile, three chunks: a single workgroup"),
])
def test_streamk_schedules_agree_with_direct_kernel(gpu, cin, cout, k, s, d, gr, act, B, L, what):
    """Every range layout of the stream-K launch (conv_mfma.hip launch_cfg / sk_u0: tile-aligned splits incl. the uneven
    two-per-tile one, whole tiles, several tiles per workgroup, the balanced split; any workgroup count) for both arithmetics
    against the scalar direct kernel, over calls that wrap the ring."""
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(cin + 13 * k + B)
    w = torch.randn(cout, cin // gr, k, generator=g) / (cin // gr * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    mods = {}
    for impl in (native.IMPL_DIRECT, native.IMPL_MFMA, native.IMPL_SPLIT16_SK):
        m = layers.CausalConv1d(cin, cout, k, s, d, gr, True, device=gpu, batch=B, max_len=L * s).load(w, bias)
        m.set_activation(act, 0.1)
        m.impl = impl
        mods[impl] = m
    for step in range(3):
        x = torch.randn(B, cin, L * s, generator=g).to(gpu)
        ref = mods[native.IMPL_DIRECT].inference(x)
        for impl in (native.IMPL_MFMA, native.IMPL_SPLIT16_SK):
            y = mods[impl].inference(x)
            assert float((y - ref).abs().max()) < 2e-5, (what, impl, step)
    assert native.device_flags() == 0


@pytest.mark.parametrize("cin,cout,k,s,d,gr,act,B,L,what", [
    (256, 256, 7, 1, 9, 1, "ELU", 1, 5, "encoder block 3 at one stream: 8 m-tiles x 10 slices"),
    (768, 768, 11, 1, 5, 3, "LeakyReLU", 1, 5, "vocoder stage 0 at one stream: 24 tiles x 15 slices of 11-12 steps"),
    (768, 768, 11, 1, 5, 3, "LeakyReLU", 6, 5, "... six streams: 30 of 32 columns"),
    (512, 64, 3, 1, 1, 1, None, 32, 1, "projector at 32 streams: 2 tiles, full 32 columns"),
    (256, 512, 10, 5, 1, 1, None, 3, 2, "strided conv (K10, stride 5): ring rows 5 apart per column"),
    (64, 512, 7, 1, 1, 1, None, 2, 1, "input conv: 28 steps, 3 slices"),
    (384, 128, 1, 1, 1, 1, None, 1, 25, "1x1 conv_out: 24 steps, 2 slices"),
    (96, 96, 3, 1, 1, 3, "LeakyReLU", 7, 4, "cout_g = 32, 6 steps: ONE slice, no exchange"),
    (1024, 128, 7, 1, 3, 1, "ELU", 2, 9, "448 steps: beyond 15 x 24 -> stays on the stream-K kernel"),
    (160, 96, 3, 1, 2, 1, "ELU", 4, 8, "cout_g = 96 (three m-tiles), 30 steps: slices of 10"),
])
def test_few_column_kernel_agrees_with_direct_kernel_and_oracle(gpu, cin, cout, k, s, d, gr, act, B, L, what):
    """conv_gv16 (csrc/conv_mfma.hip, round 5): convs of at most 32 columns run as one wave per (32-row m-tile, K slice) with the slices'
    partial sums added by the last one to arrive.  Against the scalar direct kernel and the oracle (CausalConv1d.inference,
    layers/conv_layer.py:153-156) over calls that wrap the ring, twice (bit-reproducible: the slices are added in slice order)."""
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(cin + 13 * k + B)
    w = torch.randn(cout, cin // gr, k, generator=g) / (cin // gr * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    fn = {None: lambda v: v, "ELU": torch.nn.ELU(), "LeakyReLU": torch.nn.LeakyReLU(0.1)}[act]
    mods = {}
    for key, impl in (("direct", native.IMPL_DIRECT), ("gv", native.IMPL_SPLIT16_SK), ("gv2", native.IMPL_SPLIT16_SK)):
        m = layers.CausalConv1d(cin, cout, k, s, d, gr, True, device=gpu, batch=B, max_len=L * s).load(w, bias)
        m.set_activation(act, 0.1)
        m.impl = impl
        mods[key] = m
    pad = torch.zeros(B, cin, (k - 1) * d)
    for step in range(4):
        x = torch.randn(B, cin, L * s, generator=g)
        ref, pad = O.causal_conv1d_inference(fn(x), fn(pad) if step == 0 else pad, w, bias, s, d, gr)
        yd = mods["direct"].inference(x.to(gpu))
        y1 = mods["gv"].inference(x.to(gpu))
        y2 = mods["gv2"].inference(x.to(gpu))
        want = "conv_sk16" if cin // gr * k // 16 > 360 else "conv_gv16<32>"
        assert mods["gv"].last_kernel.startswith(want), (what, mods["gv"].last_kernel)
        assert float((y1 - yd).abs().max()) < 2e-5 and float((y1.cpu() - ref).abs().max()) < 2e-5, (what, step)
        assert torch.equal(y1, y2), (what, step)
    assert native.device_flags() == 0


@pytest.mark.parametrize("cin,cout,k,s,d,gr,act,B,L,what", [
    (512, 64, 3, 1, 1, 1, None, 256, 1, "projector at 256 streams: 2 m-tiles x 8 n-tiles x 8 slices"),
    (256, 256, 7, 1, 9, 1, "ELU", 40, 5, "encoder block 3, 200 columns: the last n-tile is ragged (8 of 32)"),
    (256, 512, 10, 5, 1, 1, None, 100, 1, "strided conv at 100 streams"),
    (768, 768, 11, 1, 5, 3, "LeakyReLU", 13, 5, "vocoder stage 0, 65 columns: 24 x 3 x 15 slices = 1080 waves; one column in the last n-tile"),
    (768, 768, 11, 1, 5, 3, "LeakyReLU", 52, 5, "... 260 columns: beyond the option's limit -> the stream-K kernel"),
])
def test_few_column_kernel_over_several_column_tiles(gpu, cin, cout, k, s, d, gr, act, B, L, what):
    """conv_gv16 with option "gv16_max_columns" = 256: one wave per (m-tile, 32-column n-tile, K slice), one arrival counter per output tile.
    Same checks as above (direct kernel, oracle, bit-reproducible), over ring wrap-around."""
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(cin + 13 * k + B)
    w = torch.randn(cout, cin // gr, k, generator=g) / (cin // gr * k) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    fn = {None: lambda v: v, "ELU": torch.nn.ELU(), "LeakyReLU": torch.nn.LeakyReLU(0.1)}[act]
    native.set_option("gv16_max_columns", 256)
    try:
        mods = {}
        for key, impl in (("direct", native.IMPL_DIRECT), ("gv", native.IMPL_SPLIT16_SK), ("gv2", native.IMPL_SPLIT16_SK)):
            m = layers.CausalConv1d(cin, cout, k, s, d, gr, True, device=gpu, batch=B, max_len=L * s).load(w, bias)
            m.set_activation(act, 0.1)
            m.impl = impl
            mods[key] = m
        pad = torch.zeros(B, cin, (k - 1) * d)
        for step in range(4):
            x = torch.randn(B, cin, L * s, generator=g)
            ref, pad = O.causal_conv1d_inference(fn(x), fn(pad) if step == 0 else pad, w, bias, s, d, gr)
            yd = mods["direct"].inference(x.to(gpu))
            y1 = mods["gv"].inference(x.to(gpu))
            y2 = mods["gv2"].inference(x.to(gpu))
            assert mods["gv"].last_kernel.startswith("conv_gv16<32>" if B * L <= 256 else "conv_sk16"), (what, mods["gv"].last_kernel)
            assert float((y1 - yd).abs().max()) < 2e-5 and float((y1.cpu() - ref).abs().max()) < 2e-5, (what, step)
            assert torch.equal(y1, y2), (what, step)
        assert native.device_flags() == 0
    finally:
        native.set_option("gv16_max_columns", 32)


@pytest.mark.parametrize("cin,cout,stride,B,L", [(512, 256, 5, 1, 1), (256, 128, 5, 2, 5), (128, 64, 4, 1, 25), (64, 32, 3, 3, 10)])
def test_few_column_kernel_transposed_convs(gpu, cin, cout, stride, B, L):
    """The polyphase form of CausalConvTranspose1d.inference (layers/conv_layer.py:194-197: K = 2 * stride, two taps, stride * Cout GEMM rows
    written as `stride` ring rows per input step) through conv_gv16: upsamples.0 ... 3 of the vocoder at one to three streams."""
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(cin + stride)
    w = torch.randn(cin, cout, 2 * stride, generator=g) / (2 * cin) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    mods = {}
    for key, impl in (("direct", native.IMPL_DIRECT), ("gv", native.IMPL_SPLIT16_SK)):
        m = layers.CausalConvTranspose1d(cin, cout, 2 * stride, stride, True, device=gpu, batch=B, max_len=L).load(w, bias)
        m.set_activation("LeakyReLU", 0.1)
        m.impl = impl
        mods[key] = m
    pad = torch.zeros(B, cin, 1)
    lrelu = torch.nn.LeakyReLU(0.1)
    for step in range(4):
        x = torch.randn(B, cin, L, generator=g)
        ref, pad = O.causal_convtr1d_inference(lrelu(x), lrelu(pad) if step == 0 else pad, w, bias, stride)
        yd, y1 = mods["direct"].inference(x.to(gpu)), mods["gv"].inference(x.to(gpu))
        assert mods["gv"].last_kernel == ("conv_gv16<32>" if B * L <= 32 else mods["gv"].last_kernel)
        assert float((y1 - yd).abs().max()) < 2e-5 and float((y1.cpu() - ref).abs().max()) < 2e-5, step
    assert native.device_flags() == 0


def test_elu_of_the_kernels_against_fp64(gpu):
    """The kernels' own ELU (expm1_neg, csrc/adk_common.h) through an identity 1x1 conv on the exact-f32 kernels: relative error
    against torch's fp64 ELU <= 4e-7 over [-20, 2] (the reference applies torch.nn.ELU in fp32: layers/activation_function.py:18-22)."""
    from audiodec_amd import layers, native
    C, L = 32, 4096
    w = torch.eye(C).unsqueeze(-1)
    g = torch.Generator().manual_seed(11)
    x = torch.cat([-torch.logspace(-7, 1.3, C * L // 2, dtype=torch.float64), torch.rand(C * L // 2, generator=g, dtype=torch.float64) * 3 - 1])
    x = x[torch.randperm(C * L, generator=g)].float().view(1, C, L)
    ref = torch.nn.functional.elu(x.double())
    for impl in (native.IMPL_DIRECT, native.IMPL_MFMA):
        m = layers.CausalConv1d(C, C, 1, 1, 1, 1, False, device=gpu, batch=1, max_len=L).load(w, None)
        m.set_activation("ELU", 0.0)
        m.impl = impl
        y = m.inference(x).cpu().double()
        rel = ((y - ref).abs() / ref.abs().clamp_min(1e-30)).max().item()
        assert rel < 4e-7, (impl, rel)


def test_split16_weight_packing_kernel_matches_host_packing(gpu):
    import ctypes as ct
    from audiodec_amd import native, program
    g = torch.Generator().manual_seed(5)
    for groups, cout_g, ktot in ((3, 32, 352), (1, 64, 448), (3, 64, 704), (1, 96, 96)):
        w = torch.randn(groups * cout_g, ktot, generator=g) * 0.2
        w[0, 0], w[1, 1] = 1e-7, 3.1e-5                  # f16-subnormal hi parts
        host = program.pack_split16(w, groups)
        n = native.lib().adk_packed_weight_floats_split16(groups, cout_g, ktot)
        assert n == host.numel()
        out = torch.empty(n, device=gpu)
        wd = w.to(gpu)
        native.check(native.lib().adk_pack_weights_split16(ct.c_void_p(wd.data_ptr()), ct.c_void_p(out.data_ptr()), groups, cout_g, ktot,
                                                           native.current_stream(gpu)), "adk_pack_weights_split16")
        assert torch.equal(out.cpu().view(torch.int32), host.view(torch.int32))
    assert native.lib().adk_packed_weight_floats_split16(1, 32, 100) == -1


def test_split16_overflow_raises_device_flag(gpu):
    import ctypes as ct
    from audiodec_amd import layers, native
    m = layers.CausalConv1d(32, 32, 3, device=gpu, batch=1, max_len=64).load(torch.randn(32, 32, 3) * 0.1, torch.zeros(32))
    m.impl = native.IMPL_SPLIT16_ROWS
    flags = ct.c_int32(0)
    native.check(native.lib().adk_debug_flags(ct.byref(flags)), "flags")
    x = torch.randn(1, 32, 64)
    m.inference(x)
    native.check(native.lib().adk_debug_flags(ct.byref(flags)), "flags")
    assert flags.value == 0
    x[0, 3, 10] = 7.0e4                                  # beyond the f16 range
    for impl in (native.IMPL_SPLIT16_ROWS, native.IMPL_SPLIT16_SK):
        m.impl = impl
        m.inference(x)
        native.check(native.lib().adk_debug_flags(ct.byref(flags)), "flags")
        assert flags.value & 8, impl
        native.check(native.lib().adk_debug_flags(ct.byref(flags)), "flags")
        assert flags.value == 0                          # sticky until read, then cleared


@pytest.mark.parametrize("mode", ["lazy", "sync"])
@pytest.mark.parametrize("model", ["vctk_v1", "vctk_sym"])
def test_split16_range_overflow_is_repaired_by_the_f32_kernels(gpu, ckpt_root, model, mode):
    """The product default is the split-f16 arithmetic; its one failure mode -- an operand beyond the f16 range, |v| > 65504 --
    must not cost a direct caller anything: with guard (AudioDec's default) the step that overflowed is repeated on the
    exact-f32 kernels, in place, and the program stays on them.  Both guard modes of direct calls: "lazy" (default: the check is read
    when the result is first looked at -- here `.cpu()` --, audiodec_amd/lazy_guard.py) and "sync" (before the call returns).  Stream 1 of 3 is driven with audio scaled by 1e6 for one
    frame in the middle of a stream (the encoder's first residual block then sees activations ~1e6) and the decoder is fed a
    zq scaled by 1e5 for one frame: no exception, a RuntimeWarning each, every frame before / during / after within tolerance
    of the CPU oracle (which is exact f32 throughout), the other streams undisturbed.  Audio scaled by 1e-6 (f16 subnormal
    territory for the hi parts) needs no repair and stays within tolerance too."""
    import warnings
    from audiodec_amd import native
    B, hop, seed = 3, 300, 1337
    ad = load_audiodec(ckpt_root, model, seed, B, 1, True)
    assert ad.tx_encoder.guard and ad.decoder.guard and ad.tx_encoder.split16
    for g_ in (ad.tx_encoder, ad.rx_encoder, ad.decoder):
        g_.set_guard(True, mode)
    from audiodec_amd import lazy_guard
    tx, rx, dec = build_oracle(model, B, seed)
    audio = np.stack([synth.synth_audio(55, s, 8 * hop) for s in range(B)])
    scale_x = {2: (1, 1e6), 5: (0, 1e-6)}            # frame -> (stream, factor)
    scale_q = {3: (2, 1e5)}
    enc_prog = ad.tx_encoder._encoder()
    dec_progs = ad.decoder._decoder_stages() if hasattr(ad.decoder, "_decoder_stages") else [ad.decoder._decoder()]
    assert enc_prog.split16 and all(p.split16 for p in dec_progs)
    with torch.no_grad():
        for f in range(8):
            x = torch.from_numpy(audio[:, f * hop:(f + 1) * hop].copy())[:, None, :]
            if f in scale_x:
                x[scale_x[f][0]] *= scale_x[f][1]
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                z = ad.tx_encoder.encode(x.to(DEV))
                assert (type(z) is lazy_guard.GuardedTensor) == (mode == "lazy")
                zc = z.cpu()                                     # lazy: the first look at the result settles the log (and repairs)
            assert any(issubclass(i.category, RuntimeWarning) for i in w) == (f == 2), (f, [str(i.message) for i in w])
            oz = tx.encode(x)
            # (the stream that carried 1e6-sized samples keeps them in its state for a receptive field: its later, O(1) outputs
            # are differences of 1e6-sized f32 sums on either side -- tolerance 1e-3 there, 1e-4 everywhere else)
            dz = (zc - oz).abs().amax(dim=(1, 2)) / oz.abs().amax(dim=(1, 2)).clamp(min=1.0)      # per stream, relative to its largest value
            tol = torch.full((B,), 1e-4); tol[1] = 1e-4 if f <= 2 else 1e-3
            assert bool((dz < tol).all()), (f, dz)
            oi = tx.quantize(oz)
            zq = rx.lookup(oi)                                   # both decoders get the reference's codes
            if f in scale_q:
                zq = zq.clone(); zq[scale_q[f][0]] *= scale_q[f][1]
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                y = ad.decoder.decode(zq.to(DEV))
                yc = y.cpu()
            assert any(issubclass(i.category, RuntimeWarning) for i in w) == (f == 3), (f, [str(i.message) for i in w])
            oy = dec.decode(zq)
            dy = (yc - oy).abs().amax(dim=(1, 2)) / oy.abs().amax(dim=(1, 2)).clamp(min=1.0)
            tol = torch.full((B,), 1e-4); tol[2] = 1e-4 if f <= 3 else 1e-3
            assert bool(torch.isfinite(yc).all()) and bool((dy < tol).all()), (f, dy)
            assert enc_prog.demoted == (f >= 2) and enc_prog.split16 == (f < 2)
            assert any(p.demoted for p in dec_progs) == (f >= 3)
    assert native.device_flags() == 0
    # without the guard the failure is not lost either: it sits in the flag word until the caller's next check
    ad2 = load_audiodec(ckpt_root, model, seed, B, 1, True)
    ad2.tx_encoder.set_guard(False)
    x = torch.from_numpy(audio[:, :hop].copy())[:, None, :] * 1e6
    ad2.tx_encoder.encode(x.to(DEV))
    with pytest.raises(native.NativeError, match="f16 range"):
        native.raise_on_device_flags("test")
    assert native.device_flags() == 0


# ------------------------------------------------------------------------------------------------
# whole path against the reference's outputs
# ------------------------------------------------------------------------------------------------
def run_hip(ad, audio, chunks):
    """audio (streams, samples) or (streams, channels, samples)"""
    n = audio.shape[0]
    if audio.ndim == 2:
        audio = audio[:, None, :]
    pos, z_l, i_l, q_l, y_l = 0, [], [], [], []
    with torch.no_grad():
        for c in chunks:
            x = torch.from_numpy(np.ascontiguousarray(audio[:, :, pos:pos + c])).to(DEV)
            pos += c
            z = ad.tx_encoder.encode(x)
            idx = ad.tx_encoder.quantize(z)
            zq = ad.rx_encoder.lookup(idx)
            y = ad.decoder.decode(zq)
            if n == 1:
                idx = idx[:, None]
            z_l.append(z.cpu()); i_l.append(idx.cpu()); q_l.append(zq.cpu()); y_l.append(y.cpu())
    return (torch.cat(z_l, -1).numpy(), torch.cat(i_l, -1).numpy(), torch.cat(q_l, 1).numpy(), torch.cat(y_l, -1).numpy())


@pytest.mark.parametrize("name,max_frames", [("vctk_sym_stream", 2), ("vctk_v1_stream", 4), ("libritts_sym_file", 16),
                                             ("vctk_v2_stream", 2), ("vctk_v0_stream", 2), ("vctk_activate_sym_stream", 2),
                                             ("vctk_c16h320_sym_stream", 2), ("libritts_v1_stream", 2), ("vctk_denoise_stream", 2),
                                             ("vctk_univ_stream", 2), ("vctk_univ_sym_stream", 2),
                                             ("test_v1_noaddl_stream", 2), ("test_v0_noaddl_stream", 2), ("test_stereo_sym_stream", 2)])
@pytest.mark.parametrize("split16", [False, True], ids=["f32", "split16"])
def test_pipeline_matches_reference_fixture(gpu, golden_dir, ckpt_root, name, max_frames, split16):
    g = _load(golden_dir, name)
    model, seed, n = str(g["model"]), int(g["seed"]), int(g["n_streams"])
    chunks = golden_chunks(g)
    audio = golden_audio(g, sum(chunks))
    from audiodec_amd import native
    native.set_option("chain_min_blocks", 0 if split16 else 160)     # split16: the residual chains run as one launch even for these 1-2 streams
    try:
        ad = load_audiodec(ckpt_root, model, seed, n, max_frames, split16)
        if split16:                                              # the split kernels really are in the programs
            kinds = [ad.decoder._decoder().describe_op(i, 1) for i in range(ad.decoder._decoder().n_ops)]
            assert any(k.startswith(("conv_rl16", "conv_sk16", "conv_gv16")) for k in kinds), kinds
            assert "noaddl" in model or any(k.startswith("conv_rb16") for k in kinds), kinds     # (x + convs1(act(x)) blocks have no chain)
        z, idx, zq, y = run_hip(ad, audio, chunks)
    finally:
        native.set_option("chain_min_blocks", 0)
    assert z.shape == g["z"].shape and y.shape == g["y"].shape and idx.shape == g["idx"].shape
    assert np.abs(z - g["z"]).max() < WAVE_TOL
    explain_flips(idx, g["idx"], g["margin"], name)
    assert np.abs(zq - g["zq"]).max() < 1e-5
    assert np.abs(y - g["y"]).max() < WAVE_TOL, f"max|dy| = {np.abs(y - g['y']).max():.3e}"


@pytest.mark.parametrize("model,B,frames,split16", [("vctk_v1", 16, [1, 1, 2, 1], False), ("vctk_sym", 33, [1, 3], False),
                                                    ("vctk_v1", 16, [1, 1, 2, 1], True), ("vctk_sym", 33, [1, 3], True),
                                                    ("vctk_v1", 64, [1, 1], True)])
def test_batched_streams_match_oracle(gpu, ckpt_root, model, B, frames, split16):
    """B streams in one object == the B-stream oracle (B independent reference instances)."""
    seed = 4242
    hop = 300
    chunks = [f * hop for f in frames]
    audio = np.stack([synth.synth_audio(seed, 100 + s, sum(chunks)) for s in range(B)])
    ad = load_audiodec(ckpt_root, model, seed, B, 2, split16)
    z, idx, zq, y = run_hip(ad, audio, chunks)
    tx, rx, dec = build_oracle(model, B, seed)
    oz, oi, om, oy = [], [], [], []
    pos = 0
    with torch.no_grad():
        for c in chunks:
            x = torch.from_numpy(audio[:, pos:pos + c])[:, None, :]
            pos += c
            z_ = tx.encode(x)
            i_, m_ = tx.quantize(z_, return_margin=True)
            oy.append(dec.decode(rx.lookup(i_))); oz.append(z_); oi.append(i_); om.append(m_)
    oz = torch.cat(oz, -1).numpy(); oi = torch.cat(oi, -1).numpy(); om = torch.cat(om, -1).numpy(); oy = torch.cat(oy, -1).numpy()
    assert np.abs(z - oz).max() < WAVE_TOL
    explain_flips(idx, oi, om, f"{model} B={B}")
    assert np.abs(y - oy).max() < WAVE_TOL, f"max|dy| = {np.abs(y - oy).max():.3e}"


def test_rvq_indices_bit_exact_on_reference_latents(gpu, golden_dir):
    """Same z in -> same indices out: feed the REFERENCE's z (fixture) to the HIP quantiser."""
    from audiodec_amd import layers
    for name in ("libritts_sym_file", "vctk_v1_stream"):
        g = _load(golden_dir, name)
        _, enc_tag, _, _, _ = configs.alias(str(g["model"]))
        sd = synth.synth_state_dict(enc_tag, int(g["seed"]))
        embeds = [sd[f"quantizer.codebook.layers.{i}.embed"] for i in range(8)]
        rvq = layers.ResidualVQ(embeds, device=gpu)
        z = torch.from_numpy(g["z"]).transpose(2, 1).contiguous()          # (B, T, 64)
        _, idx = rvq.forward_index(z, flatten_idx=True)
        idx = idx.cpu().numpy()
        if idx.ndim == 2:
            idx = idx[:, None]
        explain_flips(idx, g["idx"], g["margin"], name + " (reference z)")


@pytest.mark.parametrize("rows,rows_per_wg", [(1, 2), (5, 2), (256, 0), (256, 2), (256, 4), (255, 2), (193, 4), (300, 2), (300, 0), (1021, 4)])
def test_rvq_exact_ties_pick_the_lowest_index(gpu, rows, rows_per_wg):
    """`(-dist).max(1)` returns the lowest index among equal maxima (vq_module.py:97): duplicate codes inside one wave of the
    search (64 consecutive codes), across waves, and in every stage, with rows sitting exactly on them -- on the one-row-per-
    workgroup kernel (< 192 rows, or "rvq_rows" 0 up to 256: DPP arg-max, owner-only fold), on the round-4 kernel with 2 or 4
    rows per workgroup (packed FMAs, row pairs; odd row counts leave a workgroup half empty) and on the first-round kernel
    (300 rows with "rvq_rows" 0), against the oracle."""
    from audiodec_amd import layers, native
    native.set_option("rvq_rows", rows_per_wg)
    try:
        _rvq_ties(gpu, rows)
    finally:
        native.set_option("rvq_rows", 1)


def test_rvq_row_grouping_is_bit_identical(gpu):
    """1, 2 or 4 rows per workgroup: the same indices and the same zq, bit for bit (per row the same operations in the same order)."""
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(777)
    embeds = [torch.randn(64, 1024, generator=g) * (0.7 ** i) for i in range(8)]
    rvq = layers.ResidualVQ(embeds, device=gpu)
    out = {}
    try:
        for n in (192, 256, 257, 777):
            x = torch.randn(1, n, 64, generator=g)
            for r in (0, 2, 4):
                native.set_option("rvq_rows", r)
                q, idx = rvq.forward_index(x, flatten_idx=True)
                out[r] = (q.cpu(), idx.cpu())
            for r in (2, 4):
                assert torch.equal(out[r][1], out[0][1]), (n, r)
                assert torch.equal(out[r][0], out[0][0]), (n, r, float((out[r][0] - out[0][0]).abs().max()))
    finally:
        native.set_option("rvq_rows", 1)
    assert native.device_flags() == 0


def _rvq_ties(gpu, rows):
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(4242 + rows)
    embeds = [torch.randn(64, 1024, generator=g) * (0.8 ** i) for i in range(8)]
    for st in range(8):
        a = 64 * st + 3
        embeds[st][:, a + 17] = embeds[st][:, a]           # same wave
        embeds[st][:, 1000 - st] = embeds[st][:, a]        # another wave
        embeds[st][:, 5 + st] = embeds[st][:, 900 + st]    # the duplicate with the LOWER index comes later in memory order of the pair
    x = torch.randn(1, rows, 64, generator=g)
    x[0, 0] = embeds[0][:, 3]                              # row 0 sits exactly on the three-way tie of stage 0
    if rows > 3:
        x[0, 3] = embeds[0][:, 900]                        # ... and row 3 on the (5, 900) pair
    want_q, want_idx = O.rvq_forward_index(x, embeds, flatten_idx=True)
    rvq = layers.ResidualVQ(embeds, device=gpu)
    q, idx = rvq.forward_index(x, flatten_idx=True)
    idx = idx.cpu().numpy()
    assert np.array_equal(idx, want_idx.numpy()), np.argwhere(idx != want_idx.numpy())[:5]
    assert int(idx[0, 0]) == 3 and (rows <= 3 or int(idx[0, 3]) == 5)
    assert np.abs(q.cpu().numpy() - want_q.numpy()).max() < 1e-5
    assert native.device_flags() == 0


def test_chunked_equals_one_shot_and_reset(gpu, ckpt_root):
    """Streaming invariants on the HIP path.  Chunking must not change the codes
    and may change the waveform only by fp32 round-off (the stream-K schedule splits the K sum of a
    tile at points that depend on the number of columns in the call); the same call sequence is
    bit-reproducible, and reset_buffer + warm-up reproduces the initial state exactly."""
    seed, B, hop = 99, 3, 300
    audio = np.stack([synth.synth_audio(seed, s, 8 * hop) for s in range(B)])
    ad = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 3)
    one = run_hip(ad, audio, [8 * hop])                    # split 3+3+2 internally
    ad2 = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 3)
    many = run_hip(ad2, audio, [hop, 2 * hop, hop, 3 * hop, hop])
    assert np.array_equal(one[1], many[1])
    assert np.abs(one[0] - many[0]).max() < 1e-5 and np.abs(one[3] - many[3]).max() < 1e-5
    # reset + re-warm == fresh, bit for bit (same call sequence -> same summation order)
    ad2.tx_encoder.reset_buffer(); ad2.rx_encoder.reset_buffer(); ad2.decoder.reset_buffer()
    ad2.tx_encoder.initial_encoder(8192, DEV)
    ad2.decoder.initial_decoder(ad2.rx_encoder.initial_encoder(8192, DEV))
    again = run_hip(ad2, audio, [8 * hop])
    assert np.array_equal(one[1], again[1]) and np.array_equal(one[0], again[0]) and np.array_equal(one[3], again[3])


@pytest.mark.parametrize("model,split16,stages", [("vctk_v1", False, "2"), ("vctk_v1", True, "2"), ("vctk_v0", False, "2"),
                                                  ("vctk_v1", True, "1,2,3")])
def test_two_stage_vocoder_equals_one_stage(gpu, ckpt_root, model, split16, stages):
    """set_stages(2): the vocoder as two programs (cut in front of upsample stage 2) gives bit-identical output,
    back to back or with the halves on different HIP streams (v1: grouped convs + 1x1; v0: three residual blocks
    averaged into the hand-over buffer).  A cut in front of stage 3 separates the two convs that conv_ou16 runs as one
    launch on 16 x 16 x 32 MFMAs; the two-launch form sums the same products in the same chunk order on 32 x 32 x 16:
    f32 round-off there, not bit-identity."""
    seed, B, hop = 99, 3, 300
    audio = np.stack([synth.synth_audio(seed, s, 4 * hop) for s in range(B)])
    ad1 = load_audiodec(ckpt_root, model, seed, B, 2, split16)
    os.environ["ADK_VOCODER_STAGES"] = stages
    try:
        ad2 = load_audiodec(ckpt_root, model, seed, B, 2, split16)
    finally:
        del os.environ["ADK_VOCODER_STAGES"]
    n_st = 2 if stages == "2" else len(stages.split(",")) + 1
    assert ad1.decoder.stages == 1 and ad2.decoder.stages == n_st and len(ad2.decoder._decoder_stages()) == n_st
    s2 = torch.cuda.Stream(gpu)
    splits_ou16 = split16 and "3" in stages.split(",")

    def same(a, b):
        return a.shape == b.shape and (float((a - b).abs().max()) < 2e-6 if splits_ou16 else torch.equal(a, b))
    for f0, f1 in ((0, 1), (1, 3), (3, 4)):
        x = torch.from_numpy(audio[:, f0 * hop:f1 * hop])[:, None, :].to(gpu)
        zq = ad1.rx_encoder.lookup(ad1.tx_encoder.quantize(ad1.tx_encoder.encode(x)))
        y1 = ad1.decoder.decode(zq)
        if f0 == 1:                                           # first program here, the rest on another stream, handed over by an event
            mid = ad2.decoder.decode_stage(0, zq)
            ev = torch.cuda.Event(); ev.record()
            with torch.cuda.stream(s2):
                s2.wait_event(ev)
                mid.record_stream(s2)
                for i in range(1, n_st):
                    mid = ad2.decoder.decode_stage(i, mid)
                y2 = mid
            torch.cuda.current_stream().wait_stream(s2)
        else:
            y2 = ad2.decoder.decode(zq)
        assert same(y1, y2)
    ad2.decoder.reset_stream(1)                              # per-stream reset reaches both programs
    ad1.decoder.reset_stream(1)
    x = torch.from_numpy(audio[:, :hop])[:, None, :].to(gpu)
    zq = ad1.rx_encoder.lookup(ad1.tx_encoder.quantize(ad1.tx_encoder.encode(x)))
    assert same(ad1.decoder.decode(zq), ad2.decoder.decode(zq))


def test_workgroup_share_changes_only_the_summation_order(gpu, ckpt_root):
    """adk_program_set_workgroups: fewer persistent workgroups per stream-K launch (what concurrently running programs
    use) moves the K split points, i.e. only the association of the f32 sums."""
    from audiodec_amd import native
    seed, B, hop = 7, 32, 300
    audio = np.stack([synth.synth_audio(seed, s, 2 * hop) for s in range(B)])
    ad1 = load_audiodec(ckpt_root, "vctk_v1", seed, B, 1, True)
    ad2 = load_audiodec(ckpt_root, "vctk_v1", seed, B, 1, True)
    ad2.tx_encoder.set_workgroups(64)
    ad2.decoder.set_workgroups(64)
    assert ad2.decoder._decoder().lib.adk_program_set_workgroups(ad2.decoder._decoder().h, 3) != 0     # 0 or >= 8
    for f in range(2):
        x = torch.from_numpy(audio[:, f * hop:(f + 1) * hop])[:, None, :].to(gpu)
        outs = []
        for ad in (ad1, ad2):
            idx = ad.tx_encoder.quantize(ad.tx_encoder.encode(x))
            outs.append((idx, ad.decoder.decode(ad.rx_encoder.lookup(idx))))
        assert torch.equal(outs[0][0], outs[1][0])
        assert float((outs[0][1] - outs[1][1]).abs().max()) < 1e-5
    flags = __import__("ctypes").c_int32(0)
    native.check(native.lib().adk_debug_flags(__import__("ctypes").byref(flags)), "flags")
    assert flags.value == 0


def test_transmitter_receiver_on_two_hip_streams(gpu, ckpt_root):
    """bench.py's schedule: encode+RVQ on one HIP stream, lookup+vocoder on another, codes handed over by an
    event (the reference's two streamer threads).  Concurrent stream-K kernels must not disturb each other."""
    seed, B, hop, steps = 77, 24, 300, 6
    audio = np.stack([synth.synth_audio(seed, s, steps * hop) for s in range(B)])
    ad = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 1)
    s_tx, s_rx = torch.cuda.Stream(DEV), torch.cuda.Stream(DEV)
    ys, idxs = [], []
    with torch.no_grad():
        xs = [torch.from_numpy(audio[:, i * hop:(i + 1) * hop])[:, None, :].to(DEV) for i in range(steps)]
        torch.cuda.synchronize()
        for i in range(steps):
            with torch.cuda.stream(s_tx):
                idx = ad.tx_encoder.quantize(ad.tx_encoder.encode(xs[i]))
                ev = torch.cuda.Event(); ev.record(s_tx)
            with torch.cuda.stream(s_rx):
                s_rx.wait_event(ev)
                idx.record_stream(s_rx)
                ys.append(ad.decoder.decode(ad.rx_encoder.lookup(idx))); idxs.append(idx)
        torch.cuda.synchronize()
    y = torch.cat(ys, -1).cpu().numpy(); idx = torch.cat(idxs, -1).cpu().numpy()
    tx, rx, dec = build_oracle("vctk_v1", B, 1337)
    with torch.no_grad():
        oi, om = tx.quantize(tx.encode(torch.from_numpy(audio)[:, None, :]), return_margin=True)
        oy = dec.decode(rx.lookup(oi))
    explain_flips(idx, oi.numpy(), om.numpy(), "two-stream vctk_v1")
    assert np.abs(y - oy.numpy()).max() < WAVE_TOL
    from audiodec_amd import native
    import ctypes
    f = ctypes.c_int32(0)
    native.check(native.lib().adk_debug_flags(ctypes.byref(f)), "adk_debug_flags")
    assert f.value == 0, "a stream-K workgroup timed out waiting for a partial tile"


def test_error_behaviour(gpu, ckpt_root):
    from audiodec_amd.audiodec import AudioDec, assign_model
    from audiodec_amd import native
    with pytest.raises(NotImplementedError):
        assign_model("no_such_model")
    ad = AudioDec(tx_device=DEV, rx_device=DEV)
    with pytest.raises(AssertionError):
        ad.load_transmitter("exp/does/not/exist.pkl")
    # the reference's 'cpu' defaults: mapped to the first HIP device with a warning (there is no CPU compute path)
    native._warned_cpu = False
    with pytest.warns(UserWarning, match="mapped to 'cuda:0'"):
        ad_cpu = AudioDec()
    assert ad_cpu.tx_device == "cuda:0" and ad_cpu.rx_device == "cuda:0"
    ad = load_audiodec(ckpt_root, "vctk_sym", 1337, 2, 2)
    with pytest.raises(ValueError):
        ad.tx_encoder.encode(torch.zeros(3, 1, 300, device=DEV))      # wrong stream count


# ------------------------------------------------------------------------------------------------
# "next" rows: wire format and the batched multi-stream streamer
# ------------------------------------------------------------------------------------------------
def test_wire_format_matches_oracle_bit_for_bit(gpu):
    from audiodec_amd import wire
    from oracle import wire_oracle as W
    gold_idx, gold_payload = W.make_wire_golden()
    p = wire.pack_codes(torch.from_numpy(gold_idx).to(gpu), 1024)
    assert np.array_equal(p.cpu().numpy()[0], gold_payload)
    rng = np.random.default_rng(5)
    for n_q, size, B, T in ((8, 1024, 7, 13), (16, 1024, 3, 5), (8, 512, 2, 9), (3, 1000, 4, 4)):
        bits = wire.code_bits(size)
        idx = rng.integers(0, size, (n_q, B, T)) + size * np.arange(n_q)[:, None, None]
        pay = wire.pack_codes(torch.from_numpy(idx).to(gpu), size)
        ref = W.pack(idx.reshape(n_q, B * T), bits, size).reshape(B, T, -1)
        assert pay.shape == ref.shape and np.array_equal(pay.cpu().numpy(), ref)
        back = wire.unpack_codes(pay, n_q, size).cpu().numpy().reshape(n_q, B, T)
        assert np.array_equal(back, idx)
    # edge: empty batch of frames
    assert wire.pack_codes(torch.zeros(8, 1, 0, dtype=torch.int64, device=gpu)).shape == (1, 0, 10)
    # an index that is not a code of its stage: by default (check=True) the call synchronises and raises -- a payload it returns is
    # safe to ship; the real-time tick paths pass check=False: no exception, no synchronisation, the failure sits in the sticky
    # device flags until the caller's next check (and the bad index travels as code 0, never spilling into its neighbours' bits)
    from audiodec_amd import native
    assert native.device_flags() == 0
    bad = torch.from_numpy(gold_idx).to(gpu).clone()
    bad.view(8, -1)[3, 0] = 5                                            # a stage-0 index in stage 3's row
    with pytest.raises(ValueError):
        wire.pack_codes(bad, 1024)
    assert native.device_flags() == 0                                    # (the raise consumed the flag)
    wire.pack_codes(bad, 1024, check=False)
    assert native.device_flags() == native.FLAG_BAD_CODE
    assert native.device_flags() == 0                                    # read-and-clear


def test_packed_lookup_equals_lookup(gpu, ckpt_root):
    ad = load_audiodec(ckpt_root, "vctk_sym", 1337, 4, 2)
    x = torch.from_numpy(np.stack([synth.synth_audio(3, s, 600) for s in range(4)]))[:, None, :].to(gpu)
    idx = ad.tx_encoder.quantize(ad.tx_encoder.encode(x))
    payload = ad.tx_encoder.pack(idx)
    assert payload.shape == (4, 2, 10) and payload.dtype == torch.uint8        # 80 bit per frame = 12.8 kbps @ 160 fps
    assert torch.equal(ad.rx_encoder.unpack(payload), idx)
    assert torch.equal(ad.rx_encoder.lookup_packed(payload), ad.rx_encoder.lookup(idx))


def test_batched_streamer_with_per_stream_reset(gpu, ckpt_root):
    """4 logical streams, 6 ticks of one 300-sample frame; stream 2 is reset after tick 2 and stream 3
    misses a frame at tick 4.  Oracle: one batch-1 oracle per stream, driven the same way."""
    from audiodec_amd.batched_streamer import BatchedAudioDecStreamer
    n, hop, ticks = 4, 300, 6
    ad = load_audiodec(ckpt_root, "vctk_v1", 1337, n, 1)
    st = BatchedAudioDecStreamer(ad, frame_size=hop, max_latency=10.0)
    audio = np.stack([synth.synth_audio(11, s, ticks * hop) for s in range(n)])
    oracles = [build_oracle("vctk_v1", 1, 1337) for _ in range(n)]
    outs, refs = [], []
    for t in range(ticks):
        if t == 3:
            st.reset_stream(2)
            oracles[2] = build_oracle("vctk_v1", 1, 1337)          # a fresh warmed-up instance
        frames = []
        for s in range(n):
            if s == 3 and t == 4:
                frames.append(np.zeros(hop, np.float32))             # underrun: the codec sees silence
                continue
            st.push(s, audio[s, t * hop:(t + 1) * hop])
            frames.append(audio[s, t * hop:(t + 1) * hop])
        outs.append(st.tick())
        row = []
        with torch.no_grad():
            for s in range(n):
                tx, rx, dec = oracles[s]
                x = torch.from_numpy(frames[s])[None, None, :]
                row.append(dec.decode(rx.lookup(tx.quantize(tx.encode(x))))[0, 0].numpy())
        refs.append(np.stack(row))
    out, ref = np.stack(outs), np.stack(refs)
    assert np.abs(out - ref).max() < WAVE_TOL, f"max|dy| = {np.abs(out - ref).max():.3e}"
    s_ = st.stats()
    assert s_["underruns"] == 1 and s_["frame_drops"] == 0 and abs(s_["payload_kbps_per_stream"] - 12.8) < 1e-6
