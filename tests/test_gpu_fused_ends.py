"""GPU: the two launches round 6 put at the ends of the path, each against the two-launch form it replaces (adk_set_option switches them at
run time) and against the oracle:
  * conv_oc16 (csrc/conv_oc16.hip): the vocoder's last conv_out (1x1, 96 -> 32) + LeakyReLU + output conv (K7, 32 -> 1) + tanh
    (multi_fusion.py:139-141, HiFiGAN.py:292-296) -- time slices of a stream on separate workgroups, the 6 leading columns of a slice recomputed or taken
    from the ring's history: calls of 1, 2, 3 and 5 frames (3, 5, 8, 13 slices; a ragged last slice), one stream and many, steps mixed so that
    the history of one call length feeds another;
  * conv_cin1w (csrc/conv_direct.hip): the encoder's ring write inside the launch of the Cin = 1 conv behind it -- bit-identical."""
import numpy as np
import pytest
import torch

from audiodec_amd import native, synth
from test_gpu_parity import DEV, WAVE_TOL, load_audiodec
from test_oracle_golden import build_oracle_shared_warmup, explain_flips

pytestmark = pytest.mark.gpu
HOP = 300


def _set(name, v):
    native.check(native.lib().adk_set_option(name.encode(), int(v)), "adk_set_option")


def _names(prog, frames):
    return [prog.describe_op(i, frames) for i in range(prog.n_ops)]


@pytest.mark.parametrize("B,max_frames,calls", [(3, 5, [1, 5, 2, 3, 1, 5]), (1, 2, [2, 1, 1, 2]), (64, 1, [1, 1, 1]), (37, 3, [3, 1, 2, 3])])
def test_fused_vocoder_tail_equals_the_two_launch_form_and_the_oracle(gpu, ckpt_root, B, max_frames, calls):
    model, seed = "vctk_v1", 515
    total = sum(calls) * HOP
    audio = np.stack([synth.synth_audio(seed, 11 + s % 9, total) for s in range(B)])
    try:
        _set("conv_oc16", 1)
        ad_f = load_audiodec(ckpt_root, model, seed, B, max_frames, True)
        dec = list(ad_f.decoder._decoder_stages())[-1]
        for f in set(calls):
            nm = _names(dec, f)
            assert nm.count("conv_oc16<96>") == 1 and nm[-1] == "(fused into the previous op)", (f, nm[-3:])
        ad_u = load_audiodec(ckpt_root, model, seed, B, max_frames, True)
        ys_f, ys_u, zs, idxs = [], [], [], []
        pos = 0
        with torch.no_grad():
            for f in calls:
                x = torch.from_numpy(audio[:, pos:pos + f * HOP].copy())[:, None, :].to(DEV)
                pos += f * HOP
                z = ad_f.tx_encoder.encode(x)
                idx = ad_f.tx_encoder.quantize(z)
                _set("conv_oc16", 1)
                yf = ad_f.decoder.decode(ad_f.rx_encoder.lookup(idx)).cpu()
                _set("conv_oc16", 0)
                assert "conv_oc16<96>" not in _names(dec, f)
                ad_u.tx_encoder.encode(x)
                yu = ad_u.decoder.decode(ad_u.rx_encoder.lookup(idx)).cpu()
                ys_f.append(yf); ys_u.append(yu); zs.append(z.cpu()); idxs.append(idx.cpu())
    finally:
        _set("conv_oc16", 1)
    yf, yu = torch.cat(ys_f, -1).numpy(), torch.cat(ys_u, -1).numpy()
    # the output conv is exact f32 in the order of conv_cout1_kernel; the 1x1 conv in front of it sums the chunks conv_sk16 sums: the two forms
    # agree to f32 round-off at worst (this assertion), and what they agree to is printed by -s / on failure
    d = float(np.abs(yf - yu).max())
    print(f"conv_oc16 vs conv_sk16 + conv_cout1, B={B}, calls={calls}: max|dy| = {d:.3e}")
    assert d < 2e-6, f"fused vs two launches: max|dy| = {d:.3e}"
    tx, rx, odec = build_oracle_shared_warmup(model, B, seed)
    oy, oi, om = [], [], []
    pos = 0
    with torch.no_grad():
        for f in calls:
            x = torch.from_numpy(audio[:, pos:pos + f * HOP].copy())[:, None, :]
            pos += f * HOP
            i_, m_ = tx.quantize(tx.encode(x), return_margin=True)
            oy.append(odec.decode(rx.lookup(i_))); oi.append(i_); om.append(m_)
    explain_flips(torch.cat(idxs, -1).numpy(), torch.cat(oi, -1).numpy(), torch.cat(om, -1).numpy(), f"{model} B={B}")
    oy = torch.cat(oy, -1).numpy()
    assert np.abs(yf - oy).max() < WAVE_TOL, f"max|dy| = {np.abs(yf - oy).max():.3e}"
    assert native.device_flags() == 0


@pytest.mark.parametrize("model,B,max_frames,calls", [("vctk_v1", 5, 3, [1, 3, 2, 1, 3]), ("vctk_sym", 130, 1, [1, 1, 1])])
def test_ring_write_inside_the_first_conv_is_bit_identical(gpu, ckpt_root, model, B, max_frames, calls):
    seed = 616
    total = sum(calls) * HOP
    audio = np.stack([synth.synth_audio(seed, 3 + s % 5, total) for s in range(B)])
    try:
        _set("conv_cin1w", 1)
        ad_f = load_audiodec(ckpt_root, model, seed, B, max_frames, True)
        ad_u = load_audiodec(ckpt_root, model, seed, B, max_frames, True)
        enc = ad_f.tx_encoder._encoder()
        for f in set(calls):
            assert _names(enc, f)[:2] == ["ring_write+conv_cin1", "(in the launch of the ring write)"], _names(enc, f)[:2]
        pos = 0
        with torch.no_grad():
            for k, f in enumerate(calls):
                x = torch.from_numpy(audio[:, pos:pos + f * HOP].copy())[:, None, :].to(DEV)
                pos += f * HOP
                _set("conv_cin1w", 1)
                zf = ad_f.tx_encoder.encode(x).cpu()
                _set("conv_cin1w", 0)
                assert _names(enc, f)[0] == "ring_write"
                zu = ad_u.tx_encoder.encode(x).cpu()
                assert torch.equal(zf, zu), (k, float((zf - zu).abs().max()))
    finally:
        _set("conv_cin1w", 1)
    assert native.device_flags() == 0
