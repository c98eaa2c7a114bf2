#!/usr/bin/env python3
"""Generate the parity fixtures in tests/golden/*.npz by running the UNMODIFIED reference.

Runs only in the build container (needs /root/reference).  For each case it
  1. writes the seeded synthetic checkpoints (audiodec_amd/synth.py) into a scratch root laid out
     like the reference's repo root (exp/<tag>/config.yml + checkpoint-*.pkl, stats/*.npy),
  2. imports the reference unchanged (a stub ``torchaudio`` is injected: it is imported at
     bin/stream.py:17 and models/vocoder/modules/discriminator.py:23 but never called on this
     path), loads it through its own ``AudioDec.load_transmitter / load_receiver`` on CPU and
     streams seeded audio through ``encode -> quantize -> lookup -> decode`` (demoFile.py:58-61),
     one reference instance per stream (the reference is batch-1 only),
  3. asserts the CPU oracle (oracle/audiodec_oracle.py) reproduces the reference bit-for-bit here,
  4. stores only the reference OUTPUTS (+ seeds/schedule); weights and audio are regenerated
     from the seed wherever the tests run.
"""
import os
import sys
import tempfile
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.abspath(os.path.join(HERE, "..", ".."))
REF = "/root/reference"
sys.path.insert(0, ROOT)

from audiodec_amd import configs, synth  # noqa: E402
from oracle import audiodec_oracle as O  # noqa: E402

SEED = 1337


def import_reference():
    ta = types.ModuleType("torchaudio")
    ta.functional = types.ModuleType("torchaudio.functional")
    ta.functional.spectrogram = None
    ta.save = None
    sys.modules.setdefault("torchaudio", ta)
    sys.modules.setdefault("torchaudio.functional", ta.functional)
    if REF not in sys.path:
        sys.path.insert(1, REF)
    import utils.audiodec as ref_audiodec
    return ref_audiodec


def load_reference(ref_audiodec, model):
    if model in configs.EXTRA_ALIASES:            # test models for generator options no released alias has: the reference's
        sr, enc_ckpt, dec_ckpt = configs.checkpoint_paths(model)      # loaders take any checkpoint path (bin/stream.py:56-77)
    else:
        sr, enc_ckpt, dec_ckpt = ref_audiodec.assign_model(model)
    ad = ref_audiodec.AudioDec(tx_device="cpu", rx_device="cpu")
    ad.load_transmitter(enc_ckpt)
    ad.load_receiver(enc_ckpt, dec_ckpt)
    return sr, ad


def build_oracle(model, batch):
    """Oracle twin of AudioDec.load_transmitter/load_receiver (bin/stream.py:56-77)."""
    sr, enc_tag, _, dec_tag, _ = configs.alias(model)
    mt_e, _, pe = configs.experiment(enc_tag)
    mt_d, _, pd = configs.experiment(dec_tag)
    sd_e = synth.synth_state_dict(enc_tag, SEED)
    sd_d = synth.synth_state_dict(dec_tag, SEED)
    tx = O.AutoEncoderOracle(sd_e, pe, batch)
    tx.initial_encoder(8192)
    rx = O.AutoEncoderOracle(sd_e, pe, 1)
    zq0 = rx.initial_encoder(8192)
    dec = O.build_decoder_oracle(sd_d, mt_d, pd, batch)
    dec.initial_decoder(zq0)
    return tx, rx, dec


def reference_conv_table(module):
    """Every convolution of a reference model as the reference built it: module path -> [kind, cin, cout, K, stride, dilation,
    groups, has bias, streaming history length].  Read off the reference's OWN module tree (named_modules), independently of
    audiodec_amd/arch.py -- the list the oracle and the product both enumerate their layers from."""
    table, pads = {}, {}
    for name, m in module.named_modules():
        if isinstance(m, torch.nn.ConvTranspose1d):
            table[name] = ["convT", m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.dilation[0], m.groups, m.bias is not None]
        elif isinstance(m, torch.nn.Conv1d):
            table[name] = ["conv", m.in_channels, m.out_channels, m.kernel_size[0], m.stride[0], m.dilation[0], m.groups, m.bias is not None]
        if hasattr(m, "pad_buffer"):                       # CausalConv1d / CausalConvTranspose1d (layers/conv_layer.py:141, :182)
            pads[name] = int(m.pad_buffer.shape[-1])
    for name, row in table.items():
        parent = name.rsplit(".", 1)[0]
        row.append(pads.get(parent, 0))
    return table


def arch_conv_table(specs):
    """The same table from audiodec_amd/arch.py's ConvSpec list."""
    return {s.wkey("weight")[:-len(".weight")]: ["convT" if s.kind == "convT" else "conv", s.cin, s.cout, s.k, s.stride, s.dilation,
                                                 s.groups, bool(s.bias), s.pad] for s in specs}


def check_arch(ad, model):
    """arch.py's enumeration of the convolutions == the reference's module tree, for the encoder half and the decoder half of
    `model`; returns the reference tables (stored in the fixture, so that the CPU suite can repeat the comparison without
    the reference: tests/test_oracle_golden.py::test_arch_enumeration_matches_the_reference_module_tree)."""
    from audiodec_amd import arch
    _, enc_tag, _, dec_tag, _ = configs.alias(model)
    _, _, pe = configs.experiment(enc_tag)
    mt_d, _, pd = configs.experiment(dec_tag)
    ref_enc = reference_conv_table(ad.tx_encoder)
    ref_dec = reference_conv_table(ad.decoder)
    ours_enc = arch_conv_table(arch.autoencoder_encoder_convs(pe) + arch.autoencoder_decoder_convs(pe))
    ours_dec = arch_conv_table(arch.hifigan_convs(pd) if mt_d in ("HiFiGAN", "UnivNet") else
                               arch.autoencoder_encoder_convs(pd) + arch.autoencoder_decoder_convs(pd))
    for what, ref, ours in (("encoder", ref_enc, ours_enc), ("decoder", ref_dec, ours_dec)):
        assert set(ref) == set(ours), f"{model} {what}: conv modules differ: only reference {sorted(set(ref) - set(ours))[:5]}, only arch.py {sorted(set(ours) - set(ref))[:5]}"
        for k in ref:
            assert ref[k] == ours[k], f"{model} {what} {k}: reference {ref[k]} != arch.py {ours[k]}"
    return {"encoder": ref_enc, "decoder": ref_dec}


def run_case(ref_audiodec, name, model, n_streams, schedule, one_shot_len=None):
    import json
    torch.set_num_threads(4)
    sr, ad = load_reference(ref_audiodec, model)
    conv_tables = check_arch(ad, model)
    hop = ad.get_hop_length(configs.checkpoint_paths(model)[1])
    total = one_shot_len if one_shot_len is not None else sum(schedule) * hop
    chunks = [one_shot_len] if one_shot_len is not None else [c * hop for c in schedule]
    in_ch = configs.experiment(configs.alias(model)[1])[2].get("input_channels", 1)
    # stream s = channels [s * in_ch, (s + 1) * in_ch) of the seeded audio (mono models: in_ch = 1)
    audio = np.stack([synth.synth_audio(SEED, s, total) for s in range(n_streams * in_ch)]).reshape(n_streams, in_ch, total)
    zs, idxs, zqs, ys, margins = [], [], [], [], []
    # reference: one freshly loaded instance per stream (identical warm-up state)
    for s in range(n_streams):
        inst = ad if s == 0 else load_reference(ref_audiodec, model)[1]
        z_l, i_l, q_l, y_l = [], [], [], []
        pos = 0
        with torch.no_grad():
            for c in chunks:
                x = torch.from_numpy(audio[s, :, pos:pos + c])[None]
                pos += c
                z = inst.tx_encoder.encode(x)
                idx = inst.tx_encoder.quantize(z)
                zq = inst.rx_encoder.lookup(idx)
                y = inst.decoder.decode(zq)
                z_l.append(z); i_l.append(idx); q_l.append(zq); y_l.append(y)
        zs.append(torch.cat(z_l, -1)[0]); idxs.append(torch.cat(i_l, -1))
        zqs.append(torch.cat(q_l, 1)[0]); ys.append(torch.cat(y_l, -1)[0])
    z = torch.stack(zs); idx = torch.stack(idxs, 1); zq = torch.stack(zqs); y = torch.stack(ys)
    # oracle: one batch-1 instance per stream must agree bit-for-bit with the reference in this
    # container; one batched instance (what the GPU parity tests use as the B-stream oracle) must
    # agree to fp32 round-off (ATen picks a different conv kernel for batch > 1)
    def run_oracle(streams):
        tx, rx, dec = build_oracle(model, len(streams))
        oz, oi, oq, oy, om = [], [], [], [], []
        pos = 0
        with torch.no_grad():
            for c in chunks:
                x = torch.from_numpy(audio[streams, :, pos:pos + c])
                pos += c
                z_ = tx.encode(x)
                i_, m_ = tx.quantize(z_, return_margin=True)
                if len(streams) == 1:
                    i_, m_ = i_[:, None], m_[:, None]
                q_ = rx.lookup(i_)
                y_ = dec.decode(q_)
                oz.append(z_); oi.append(i_); oq.append(q_); oy.append(y_); om.append(m_)
        return torch.cat(oz, -1), torch.cat(oi, -1), torch.cat(oq, 1), torch.cat(oy, -1), torch.cat(om, -1)

    per = [run_oracle([s]) for s in range(n_streams)]
    oz, oi, oq, oy, om = (torch.cat([p[k] for p in per], 1 if k in (1, 4) else 0) for k in range(5))
    assert torch.equal(oi, idx), f"{name}: oracle indices differ from the reference"
    for a, b, what in ((oz, z, "z"), (oq, zq, "zq"), (oy, y, "y")):
        d = float((a - b).abs().max())
        assert d == 0.0, f"{name}: oracle {what} differs from the reference by {d}"
    if n_streams > 1:
        bz, bi, bq, by, _ = run_oracle(list(range(n_streams)))
        assert torch.equal(bi, idx), f"{name}: batched oracle indices differ from the reference"
        print(f"  batched oracle vs reference: max|dz| {float((bz - z).abs().max()):.2e} "
              f"max|dy| {float((by - y).abs().max()):.2e}")
        assert float((by - y).abs().max()) < 2e-5
    out = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(
        out, model=model, seed=SEED, n_streams=n_streams, hop=hop, sample_rate=sr,
        schedule=np.asarray(schedule if one_shot_len is None else [], np.int64),
        one_shot_len=-1 if one_shot_len is None else one_shot_len,
        z=z.numpy(), idx=idx.numpy(), zq=zq.numpy(), y=y.numpy(), margin=om.numpy(),
        ref_convs=np.asarray(json.dumps(conv_tables, sort_keys=True)))
    print(f"{name}: frames {z.shape[-1]} z std {float(z.std()):.3f} |y|max {float(y.abs().max()):.3f} "
          f"min top-2 margin {float(om.min()):.3e} distinct idx {len(torch.unique(idx))} "
          f"-> {os.path.getsize(out)} B  (oracle == reference: exact)")


def op_cases(ref_audiodec):
    """Layer-level known answers from the reference's own layer classes (inputs: op_cases.py)."""
    from layers.conv_layer import CausalConv1d, CausalConvTranspose1d
    from layers.vq_module import ResidualVQ
    sys.path.insert(0, HERE)
    import op_cases as C
    out = {}
    for n, (ci, co, k, s, d, gr, b, L1, L2) in enumerate(C.CONVS):
        x1, x2, w, bias = C.conv_inputs(n)
        m = CausalConv1d(ci, co, k, s, d, gr, b).eval()
        with torch.no_grad():
            m.conv.weight.copy_(w)
            if b:
                m.conv.bias.copy_(bias)
            y1 = m.inference(x1); y2 = m.inference(x2)
            o1, p1 = O.causal_conv1d_inference(x1, torch.zeros(1, ci, (k - 1) * d), w, bias, s, d, gr)
            o2, p2 = O.causal_conv1d_inference(x2, p1, w, bias, s, d, gr)
        assert torch.equal(o1, y1) and torch.equal(o2, y2) and torch.equal(p2, m.pad_buffer)
        out[f"conv{n}_y1"], out[f"conv{n}_y2"] = y1.numpy(), y2.numpy()
        out[f"conv{n}_pad"] = m.pad_buffer.numpy()
    for n, (ci, co, s, L1, L2) in enumerate(C.CONVTS):
        x1, x2, w, bias = C.convt_inputs(n)
        m = CausalConvTranspose1d(ci, co, 2 * s, s).eval()
        with torch.no_grad():
            m.deconv.weight.copy_(w); m.deconv.bias.copy_(bias)
            y1 = m.inference(x1); y2 = m.inference(x2)
            o1, p1 = O.causal_convtr1d_inference(x1, torch.zeros(1, ci, 1), w, bias, s)
            o2, p2 = O.causal_convtr1d_inference(x2, p1, w, bias, s)
        assert torch.equal(o1, y1) and torch.equal(o2, y2) and torch.equal(p2, m.pad_buffer)
        out[f"convT{n}_y1"], out[f"convT{n}_y2"] = y1.numpy(), y2.numpy()
        out[f"convT{n}_pad"] = m.pad_buffer.numpy()
    # residual VQ incl. an engineered exact tie (lowest index must win, vq_module.py:98)
    embeds, x = C.rvq_inputs()
    rvq = ResidualVQ(dim=64, num_quantizers=C.RVQ_STAGES, codebook_size=1024).eval()
    for l, e in zip(rvq.layers, embeds):
        l.embed.copy_(e)
    rvq.initial()
    with torch.no_grad():
        q, idx = rvq.forward_index(x.clone(), flatten_idx=True)
        zq = rvq.lookup(idx)
    oq, oi = O.rvq_forward_index(x, embeds, True)
    assert torch.equal(oi, idx) and torch.equal(oq, q)
    assert torch.equal(O.rvq_lookup(idx, O.rvq_codebook(embeds)), zq)
    assert int(idx[0, 7]) == 123
    out["rvq_idx"] = idx.numpy(); out["rvq_q"] = q.numpy(); out["rvq_zq"] = zq.numpy()
    path = os.path.join(HERE, "ops.npz")
    np.savez_compressed(path, **out)
    print("ops:", len(out), "arrays ->", os.path.getsize(path), "B (oracle == reference: exact)")


def offline_case(name, model, lengths):
    """Known answers for the file-level drivers (codecTest.py:78-95, codecStatistic.py:92-113): the reference's
    NON-streaming Generator classes run ``encoder -> projector -> quantizer`` and ``decoder`` / the vocoder
    forward on whole (ragged-length) utterances; StandardScaler over the code vectors gives the stats."""
    from models.autoencoder.AudioDec import Generator as ref_audiodec_generator
    from models.vocoder.HiFiGAN import Generator as ref_hifigan_generator
    from sklearn.preprocessing import StandardScaler
    torch.set_num_threads(4)
    sr, enc_tag, _, dec_tag, _ = configs.alias(model)
    _, _, pe = configs.experiment(enc_tag)
    mt_d, _, pd = configs.experiment(dec_tag)
    sd_e, sd_d = synth.synth_state_dict(enc_tag, SEED), synth.synth_state_dict(dec_tag, SEED)
    enc = ref_audiodec_generator(**pe)
    enc.load_state_dict(sd_e)
    enc.eval()
    if mt_d in ("HiFiGAN", "UnivNet"):
        dec = ref_hifigan_generator(**pd)
    else:
        dec = ref_audiodec_generator(**pd)
    dec.load_state_dict(sd_d)
    dec.eval()
    o_enc = O.AutoEncoderOracle(sd_e, pe, 1)
    o_dec = O.build_decoder_oracle(sd_d, mt_d, pd, 1)
    out = dict(model=model, seed=SEED, lengths=np.asarray(lengths, np.int64), sample_rate=sr)
    scaler = StandardScaler()
    with torch.no_grad():
        for n, L in enumerate(lengths):
            audio = synth.synth_audio(SEED, 100 + n, L)[:, None].astype(np.float64)       # (T, C=1) like sf.read
            x = torch.tensor(audio, dtype=torch.float).transpose(1, 0).unsqueeze(1)        # codecTest.py:80-83
            zq, _, _ = enc.quantizer(enc.projector(enc.encoder(x)))
            y = dec(zq) if mt_d in ("HiFiGAN", "UnivNet") else dec.decoder(zq)
            ozq = o_enc.analyze(x)
            oy = o_dec.synthesize(ozq)
            assert torch.equal(ozq, zq), f"{name}: oracle zq differs from the reference forward"
            assert torch.equal(oy, y), f"{name}: oracle y differs by {float((oy - y).abs().max())}"
            scaler.partial_fit(zq.squeeze(0).transpose(1, 0).numpy())
            out[f"zq{n}"], out[f"y{n}"] = zq.numpy(), y.numpy()
    out["stats"] = np.stack([scaler.mean_, scaler.scale_], axis=0).astype(np.float32)
    path = os.path.join(HERE, f"{name}.npz")
    np.savez_compressed(path, **out)
    print(f"{name}: {len(lengths)} utterances, |y|max {max(float(np.abs(out[f'y{n}']).max()) for n in range(len(lengths))):.3f} "
          f"-> {os.path.getsize(path)} B  (oracle == reference: exact)")


OFFLINE = {
    "vctk_v1_offline": ("vctk_v1", [2000, 1234]),
    "vctk_sym_offline": ("vctk_sym", [1500, 901]),
}

# 65 frames in 29 calls: mostly single frames (the streaming steady state: the fused residual-chain kernels), with longer calls
# in between (which the product chops into max_frames pieces and, beyond one frame, runs op by op) -- every model sees both
# paths alternate on the same state
LONG = [1, 2, 1, 3, 1, 1, 4, 1, 1, 2, 1, 1, 8, 1, 1, 1, 3, 1, 1, 1, 2, 1, 1, 5, 1, 1, 1, 16, 1]

CASES = {
    # name: (model alias, n_streams, chunk schedule in frames, one-shot length in samples)
    "vctk_sym_stream": ("vctk_sym", 2, LONG, None),
    "vctk_v1_stream": ("vctk_v1", 2, LONG, None),
    "libritts_sym_file": ("libritts_sym", 1, None, 24000),          # BASELINE config 1 (demoFile)
    "vctk_v0_stream": ("vctk_v0", 1, LONG, None),
    "vctk_v2_stream": ("vctk_v2", 1, LONG, None),
    "vctk_activate_sym_stream": ("vctk_activate_sym", 1, LONG, None),
    "vctk_c16h320_sym_stream": ("vctk_c16h320_sym", 1, LONG, None),
    # the remaining aliases of utils/audiodec.py:109-179 (round 2)
    "libritts_v1_stream": ("libritts_v1", 1, LONG, None),
    "vctk_denoise_stream": ("vctk_denoise", 1, LONG, None),
    "vctk_univ_stream": ("vctk_univ", 1, LONG, None),
    "vctk_univ_sym_stream": ("vctk_univ_sym", 1, LONG, None),
    # HiFiGANResidualBlock(use_additional_convs=False) (residual_block.py:100-105), grouped (v1-shaped) and MRF (v0-shaped)
    "test_v1_noaddl_stream": ("test_v1_noaddl", 2, LONG, None),
    "test_v0_noaddl_stream": ("test_v0_noaddl", 1, LONG, None),
    # input_channels = output_channels = 2 (AudioDec.py:229-231; no released config, the generator takes the parameters)
    "test_stereo_sym_stream": ("test_stereo_sym", 2, LONG, None),
}


def main(argv):
    names = argv[1:] or (["ops"] + list(CASES) + list(OFFLINE))
    ref_audiodec = import_reference()
    with tempfile.TemporaryDirectory() as root:
        os.chdir(root)
        written = set()
        for name in names:
            if name == "ops":
                op_cases(ref_audiodec)
                continue
            if name in OFFLINE:
                model, lengths = OFFLINE[name]
                if model not in written:
                    synth.write_model(root, model, SEED)       # the vocoder Generator reads stats/*.npy (HiFiGAN.py:126-131)
                    written.add(model)
                offline_case(name, model, lengths)
                continue
            model, n, sched, one = CASES[name]
            if model not in written:
                synth.write_model(root, model, SEED)
                written.add(model)
            run_case(ref_audiodec, name, model, n, sched, one)
        os.chdir(HERE)


if __name__ == "__main__":
    main(sys.argv)
