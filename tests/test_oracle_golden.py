"""CPU: the oracle (oracle/audiodec_oracle.py) against the committed reference outputs.

The fixtures were produced by the unmodified reference in the build container
(tests/golden/make_golden.py, where oracle == reference bit-for-bit was asserted).  On another
host CPU ATen may pick different conv kernels, so waveforms are compared to fp32 round-off and
indices exactly (a flip is reported with the reference's own top-2 margin).
"""
import os

import numpy as np
import pytest
import torch

from audiodec_amd import configs, synth
from oracle import audiodec_oracle as O
import op_cases as C

TOL = 2e-5


def _load(golden_dir, name):
    return np.load(os.path.join(golden_dir, f"{name}.npz"), allow_pickle=False)


def build_oracle(model, batch, seed):
    sr, enc_tag, _, dec_tag, _ = configs.alias(model)
    mt_e, _, pe = configs.experiment(enc_tag)
    mt_d, _, pd = configs.experiment(dec_tag)
    sd_e = synth.synth_state_dict(enc_tag, seed)
    sd_d = synth.synth_state_dict(dec_tag, seed)
    tx = O.AutoEncoderOracle(sd_e, pe, batch)
    tx.initial_encoder(8192)
    rx = O.AutoEncoderOracle(sd_e, pe, 1)
    zq0 = rx.initial_encoder(8192)
    dec = O.build_decoder_oracle(sd_d, mt_d, pd, batch)
    dec.initial_decoder(zq0)
    return tx, rx, dec


def widen_oracle(o, batch):
    """Carry `batch` streams in an oracle that was warmed up with one: every stream saw the same silence, so the
    per-layer pad buffers are replicated.  (ATen's CPU convs are not bit-invariant to the batch size -- a batch-3
    warm-up differs from a batch-1 one by f32 round-off, ~1e-7 -- so this equals a `batch`-stream warm-up to round-off,
    and equals `batch` separate reference instances exactly at the moment of widening; checked below on the CPU.)"""
    o.batch = batch
    for k in list(o.pad):
        o.pad[k] = o.pad[k].expand(batch, -1, -1).clone()
    return o


def build_oracle_shared_warmup(model, batch, seed):
    """build_oracle for many streams at the cost of ONE warm-up (the 256-stream parity tests)."""
    tx, rx, dec = build_oracle(model, 1, seed)
    return widen_oracle(tx, batch), rx, widen_oracle(dec, batch)


def golden_audio(g, total):
    """(n_streams, in_ch, total): stream s = channels [s * in_ch, (s + 1) * in_ch) of the seeded audio (mono models: in_ch = 1),
    exactly as tests/golden/make_golden.py fed the reference."""
    model, seed, n = str(g["model"]), int(g["seed"]), int(g["n_streams"])
    in_ch = configs.experiment(configs.alias(model)[1])[2].get("input_channels", 1)
    return np.stack([synth.synth_audio(seed, s, total) for s in range(n * in_ch)]).reshape(n, in_ch, total)


def golden_chunks(g):
    hop = int(g["hop"])
    if int(g["one_shot_len"]) > 0:
        return [int(g["one_shot_len"])]
    return [int(c) * hop for c in g["schedule"]]


def explain_flips(idx, ref_idx, margin, what):
    """Indices must match; any mismatch is reported with the reference's top-2 margin."""
    if np.array_equal(idx, ref_idx):
        return
    bad = np.argwhere(idx != ref_idx)
    first = {}
    for q, b, t in bad:                      # only the first flipped stage of a frame is a decision
        first.setdefault((b, t), q)
    msg = [f"{what}: {len(first)} frame(s) with flipped RVQ indices"]
    for (b, t), q in first.items():
        msg.append(f"  stream {b} frame {t} stage {q}: got {idx[q, b, t]} ref {ref_idx[q, b, t]} ref margin {margin[q, b, t]:.3e}")
    raise AssertionError("\n".join(msg))


CASES = ["vctk_sym_stream", "vctk_v1_stream", "libritts_sym_file", "vctk_v0_stream", "vctk_v2_stream",
         "vctk_activate_sym_stream", "vctk_c16h320_sym_stream", "libritts_v1_stream", "vctk_denoise_stream",
         "vctk_univ_stream", "vctk_univ_sym_stream",          # all 11 aliases of utils/audiodec.py:109-179
         "test_v1_noaddl_stream", "test_v0_noaddl_stream",    # + use_additional_convs=False (configs.EXTRA_ALIASES)
         "test_stereo_sym_stream"]                            # + input_channels = output_channels = 2


@pytest.mark.parametrize("name", CASES)
def test_oracle_matches_reference_fixture(golden_dir, name):
    torch.set_num_threads(4)
    g = _load(golden_dir, name)
    model, seed, n = str(g["model"]), int(g["seed"]), int(g["n_streams"])
    chunks = golden_chunks(g)
    audio = golden_audio(g, sum(chunks))
    zs, idxs, ys = [], [], []
    for s in range(n):                                   # B streams = B batch-1 instances
        tx, rx, dec = build_oracle(model, 1, seed)
        pos, z_l, i_l, y_l = 0, [], [], []
        with torch.no_grad():
            for c in chunks:
                x = torch.from_numpy(audio[s:s + 1, :, pos:pos + c])
                pos += c
                z = tx.encode(x)
                idx = tx.quantize(z)
                y = dec.decode(rx.lookup(idx))
                z_l.append(z); i_l.append(idx); y_l.append(y)
        zs.append(torch.cat(z_l, -1)[0]); idxs.append(torch.cat(i_l, -1)); ys.append(torch.cat(y_l, -1)[0])
    z = torch.stack(zs).numpy(); idx = torch.stack(idxs, 1).numpy(); y = torch.stack(ys).numpy()
    assert np.abs(z - g["z"]).max() < TOL
    explain_flips(idx, g["idx"], g["margin"], name)
    assert np.abs(y - g["y"]).max() < TOL


@pytest.mark.parametrize("name", [c for c in CASES])
def test_arch_enumeration_matches_the_reference_module_tree(golden_dir, name):
    """audiodec_amd/arch.py is the list of convolutions BOTH the product and the oracle enumerate their layers from.  The
    fixtures carry the same table read off the reference's own module tree (named_modules() of the loaded reference models:
    kind, channels, K, stride, dilation, groups, bias, streaming history length -- tests/golden/make_golden.py, where the
    comparison is also asserted against the live reference): an enumeration mistake shared by product and oracle cannot hide
    behind outputs that happen to agree."""
    import json
    from audiodec_amd import arch
    g = _load(golden_dir, name)
    ref = json.loads(str(g["ref_convs"]))
    model = str(g["model"])
    _, enc_tag, _, dec_tag, _ = configs.alias(model)
    _, _, pe = configs.experiment(enc_tag)
    mt_d, _, pd = configs.experiment(dec_tag)

    def table(specs):
        return {s.wkey("weight")[:-len(".weight")]: ["convT" if s.kind == "convT" else "conv", s.cin, s.cout, s.k, s.stride, s.dilation,
                                                     s.groups, bool(s.bias), s.pad] for s in specs}
    ours = {"encoder": table(arch.autoencoder_encoder_convs(pe) + arch.autoencoder_decoder_convs(pe)),
            "decoder": table(arch.hifigan_convs(pd) if mt_d in ("HiFiGAN", "UnivNet") else
                             arch.autoencoder_encoder_convs(pd) + arch.autoencoder_decoder_convs(pd))}
    for half in ("encoder", "decoder"):
        assert len(ref[half]) >= 20
        assert set(ref[half]) == set(ours[half]), (half, sorted(set(ref[half]) ^ set(ours[half]))[:6])
        for k, row in ref[half].items():
            assert row == ours[half][k], (half, k, row, ours[half][k])


def test_stream_fixtures_carry_volume(golden_dir):
    """Every streaming fixture holds >= 64 frames per stream (>= 512 RVQ decisions, hundreds of distinct codes), single-frame and
    multi-frame calls mixed."""
    for name in CASES:
        g = _load(golden_dir, name)
        if int(g["one_shot_len"]) > 0:
            continue
        sched = [int(c) for c in g["schedule"]]
        assert sum(sched) >= 64 and sched.count(1) >= 16 and max(sched) >= 8, (name, sched)
        assert g["idx"].shape[-1] == sum(sched) and len(np.unique(g["idx"])) >= 400, (name, len(np.unique(g["idx"])))


def test_oracle_layers_match_reference_fixture(golden_dir):
    g = _load(golden_dir, "ops")
    for n, (ci, co, k, s, d, gr, b, L1, L2) in enumerate(C.CONVS):
        x1, x2, w, bias = C.conv_inputs(n)
        o1, p1 = O.causal_conv1d_inference(x1, torch.zeros(1, ci, (k - 1) * d), w, bias, s, d, gr)
        o2, p2 = O.causal_conv1d_inference(x2, p1, w, bias, s, d, gr)
        assert np.abs(o1.numpy() - g[f"conv{n}_y1"]).max() < 1e-5
        assert np.abs(o2.numpy() - g[f"conv{n}_y2"]).max() < 1e-5
        assert np.array_equal(p2.numpy(), g[f"conv{n}_pad"])
    for n, (ci, co, s, L1, L2) in enumerate(C.CONVTS):
        x1, x2, w, bias = C.convt_inputs(n)
        o1, p1 = O.causal_convtr1d_inference(x1, torch.zeros(1, ci, 1), w, bias, s)
        o2, p2 = O.causal_convtr1d_inference(x2, p1, w, bias, s)
        assert np.abs(o1.numpy() - g[f"convT{n}_y1"]).max() < 1e-5
        assert np.abs(o2.numpy() - g[f"convT{n}_y2"]).max() < 1e-5
        assert np.array_equal(p2.numpy(), g[f"convT{n}_pad"])
    embeds, x = C.rvq_inputs()
    q, idx = O.rvq_forward_index(x, embeds, True)
    assert np.array_equal(idx.numpy(), g["rvq_idx"])
    assert int(idx[0, 7]) == 123                          # exact tie -> lowest index (vq_module.py:98)
    assert np.abs(q.numpy() - g["rvq_q"]).max() < 1e-5
    zq = O.rvq_lookup(idx, O.rvq_codebook(embeds))
    assert np.abs(zq.numpy() - g["rvq_zq"]).max() < 1e-5


def test_shared_warmup_equals_per_stream_warmup():
    a = build_oracle("vctk_v1", 3, 1337)
    b = build_oracle_shared_warmup("vctk_v1", 3, 1337)
    for oa, ob in ((a[0], b[0]), (a[2], b[2])):
        assert oa.pad.keys() == ob.pad.keys()
        for k in oa.pad:
            assert oa.pad[k].shape == ob.pad[k].shape and float((oa.pad[k] - ob.pad[k]).abs().max()) < 1e-5, k
    x = torch.from_numpy(np.stack([synth.synth_audio(5, s, 300) for s in range(3)]))[:, None, :]
    with torch.no_grad():
        ia = a[0].quantize(a[0].encode(x)); ib = b[0].quantize(b[0].encode(x))
        assert torch.equal(ia, ib)
        assert float((a[2].decode(a[1].lookup(ia)) - b[2].decode(b[1].lookup(ib))).abs().max()) < TOL


# ---- the reference's implied invariants, held by the oracle -------------
def test_chunked_streaming_equals_one_shot():
    tx, rx, dec = build_oracle("vctk_sym", 1, 1337)
    tx2, rx2, dec2 = build_oracle("vctk_sym", 1, 1337)
    x = torch.from_numpy(synth.synth_audio(7, 0, 3000))[None, None, :]
    with torch.no_grad():
        z1 = tx.encode(x); i1 = tx.quantize(z1); y1 = dec.decode(rx.lookup(i1))
        zs, is_, ys = [], [], []
        for c in range(0, 3000, 300):
            z = tx2.encode(x[:, :, c:c + 300]); i = tx2.quantize(z)
            zs.append(z); is_.append(i); ys.append(dec2.decode(rx2.lookup(i)))
    assert torch.equal(torch.cat(is_, -1), i1)
    assert float((torch.cat(ys, -1) - y1).abs().max()) < 1e-5


def test_forward_equals_streaming_after_receptive_field():
    _, enc_tag, _, _, _ = configs.alias("vctk_sym")
    _, _, p = configs.experiment(enc_tag)
    sd = synth.synth_state_dict(enc_tag, 1337)
    a = O.AutoEncoderOracle(sd, p, 1)
    b = O.AutoEncoderOracle(sd, p, 1)
    b.initial_encoder(8192)
    x = torch.from_numpy(synth.synth_audio(9, 0, 12000))[None, None, :]
    with torch.no_grad():
        zf = a.encode(x, streaming=False)
        zs = b.encode(x)
    skip = 8192 // 300 + 1
    assert float((zf[..., skip:] - zs[..., skip:]).abs().max()) < 1e-5
