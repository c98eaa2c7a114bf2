"""Seeded synthetic checkpoints in the reference's on-disk format.

The reference's trained ``.pkl`` files are a GitHub-release download (README.md:63,141) and there
is no network, so tests, ``bench.py`` and the parity fixtures use *random-init weights of the
same architecture* written exactly where/how the reference expects them:

    <root>/exp/<tag>/config.yml                       (model_type, sampling_rate, generator_params)
    <root>/exp/<tag>/checkpoint-<N>steps.pkl          torch.save({'model': {'generator': sd}})
    <root>/stats/<name>.npy                           (2, 64) float32 mean / scale

(state-dict layout: trainer/trainerGAN.py:95-121; keys as in SURVEY.md Appendix A.)  The same
files feed the unmodified reference (``tests/golden/make_golden.py``) and this package's loader.

Determinism: every tensor is ``numpy.random.Generator(PCG64(seed, key))`` output times a constant
from ``synth_calib.json`` (per-conv weight std, per-stage codebook mean/std).  The constants were
produced once by ``tests/golden/make_calib.py`` (LSUV-style pass so activations stay O(1) through
the 30..78 conv layers and each RVQ stage uses hundreds of distinct codes -- the default inits give
one code per stage) and are frozen, so the build container and the GPU box
generate bit-identical weights without running any data-dependent calibration.
"""
import json
import os
import zlib

import numpy as np
import torch
import yaml

from . import arch, configs

_CALIB_PATH = os.path.join(os.path.dirname(__file__), "synth_calib.json")
_CALIB = None

# tags that share an architecture share calibration constants
_CALIB_ALIAS = {
    "autoencoder/symAD_libritts_24000_hop300": "autoencoder/symAD_vctk_48000_hop300",
    "autoencoder/symADuniv_vctk_48000_hop300": "autoencoder/symAD_vctk_48000_hop300",
    "denoise/symAD_vctk_48000_hop300": "autoencoder/symAD_vctk_48000_hop300",
    "vocoder/AudioDec_v1_symAD_libritts_24000_hop300_clean": "vocoder/AudioDec_v1_symAD_vctk_48000_hop300_clean",
    "vocoder/AudioDec_v3_symADuniv_vctk_48000_hop300_clean": "vocoder/AudioDec_v1_symAD_vctk_48000_hop300_clean",
    "autoencoder/test_stereo_symAD_vctk_48000_hop300": "autoencoder/symAD_vctk_48000_hop300",
    "vocoder/test_v1_noaddl_symAD_vctk_48000_hop300": "vocoder/AudioDec_v1_symAD_vctk_48000_hop300_clean",
    "vocoder/test_v0_noaddl_symAD_vctk_48000_hop300": "vocoder/AudioDec_v0_symAD_vctk_48000_hop300_clean",
}

CODEBOOK_SPREAD = 0.44   # sqrt(1 - 2^(-2*10/64)): rate-distortion scale for 10 bit / 64 dim


def calib_for(tag):
    global _CALIB
    if _CALIB is None:
        if os.path.exists(_CALIB_PATH):
            with open(_CALIB_PATH) as f:
                _CALIB = json.load(f)
        else:
            _CALIB = {}
    return _CALIB.get(_CALIB_ALIAS.get(tag, tag), None)


def _rng(seed, *key):
    h = zlib.crc32("/".join(str(k) for k in key).encode())
    return np.random.Generator(np.random.PCG64([int(seed), h]))


def _randn(seed, key, shape):
    return _rng(seed, *key).standard_normal(shape).astype(np.float32)


def default_std(spec):
    fan_in = (spec.cin // spec.groups) * spec.k
    if spec.kind == "convT":
        fan_in = spec.cin * 2            # two taps contribute to every output sample
    return float(np.float32(1.0 / np.sqrt(fan_in)))


def convs_for(model_type, params):
    if model_type in ("symAudioDec", "symAudioDecUniv"):
        return arch.autoencoder_encoder_convs(params) + arch.autoencoder_decoder_convs(params)
    return arch.hifigan_convs(params)


def synth_stats(tag, seed):
    """(2, 64) float32 [mean; scale] for the vocoder input normalisation (HiFiGAN.py:206-219)."""
    mean = (0.1 * _randn(seed, (tag, "mean"), (64,))).astype(np.float32)
    scale = (0.7 + 0.7 * _rng(seed, tag, "scale").random(64)).astype(np.float32)
    return np.stack([mean, scale]).astype(np.float32)


def synth_state_dict(tag, seed=1337, calib="auto", std_override=None):
    """Reference-format generator state dict (dict of CPU float32 torch tensors)."""
    model_type, _, params = configs.experiment(tag)
    cal = calib_for(tag) if calib == "auto" else calib
    stds = dict((cal or {}).get("std", {}))
    if std_override:
        stds.update(std_override)
    sd = {}
    for s in convs_for(model_type, params):
        std = np.float32(stds.get(s.name, default_std(s)))
        w = _randn(seed, (tag, s.name, "w"), s.wshape) * std
        if s.wn:
            # effective weight = g * v / ||v||  (torch._weight_norm, dim 0); g = ||v|| * u exercises the fold
            norm = np.sqrt((w.astype(np.float32) ** 2).reshape(w.shape[0], -1).sum(1, dtype=np.float32))
            u = (0.8 + 0.45 * _rng(seed, tag, s.name, "g").random(w.shape[0])).astype(np.float32)
            sd[s.wkey("weight_g")] = torch.from_numpy((norm * u).astype(np.float32).reshape(-1, 1, 1))
            sd[s.wkey("weight_v")] = torch.from_numpy(w)
        else:
            sd[s.wkey("weight")] = torch.from_numpy(w)
        if s.bias:
            nb = s.cout
            sd[s.wkey("bias")] = torch.from_numpy(0.1 * _randn(seed, (tag, s.name, "b"), (nb,)))
        if s.kind != "conv1x1":
            sd[f"{s.name}.pad_buffer"] = torch.zeros(1, s.cin, s.pad)
    if model_type in ("symAudioDec", "symAudioDecUniv"):
        n_q, dim, size = params["codebook_num"], params["code_dim"], params["codebook_size"]
        for i in range(n_q):
            e = _randn(seed, (tag, "embed", i), (dim, size))
            if cal and "cb_mu" in cal:
                mu = np.asarray(cal["cb_mu"][i], np.float32)[:, None]
                sg = np.asarray(cal["cb_sigma"][i], np.float32)[:, None]
                e = (mu + np.float32(CODEBOOK_SPREAD) * sg * e).astype(np.float32)
            pre = f"quantizer.codebook.layers.{i}"
            sd[f"{pre}.embed"] = torch.from_numpy(e)
            sd[f"{pre}.cluster_size"] = torch.zeros(size)
            sd[f"{pre}.embed_avg"] = torch.from_numpy(e.copy())
    else:
        st = synth_stats(tag, seed)
        sd["mean"] = torch.from_numpy(st[0].copy())
        sd["scale"] = torch.from_numpy(st[1].copy())
    return sd


def write_experiment(root, tag, steps, seed=1337, sd=None):
    """Write config.yml + checkpoint-<steps>steps.pkl (+ stats .npy) under ``root``; returns ckpt path."""
    model_type, sr, params = configs.experiment(tag)
    d = os.path.join(root, "exp", *tag.split("/"))
    os.makedirs(d, exist_ok=True)
    with open(os.path.join(d, "config.yml"), "w") as f:
        yaml.safe_dump({"model_type": model_type, "sampling_rate": sr, "generator_params": params}, f)
    if sd is None:
        sd = synth_state_dict(tag, seed)
    ckpt = os.path.join(d, f"checkpoint-{steps}steps.pkl")
    torch.save({"model": {"generator": sd}}, ckpt)
    if "stats" in params and params["stats"]:
        sp = os.path.join(root, params["stats"])
        os.makedirs(os.path.dirname(sp), exist_ok=True)
        np.save(sp, np.stack([sd["mean"].numpy(), sd["scale"].numpy()]).astype(np.float32))
    return ckpt


def write_model(root, model, seed=1337):
    """Write both checkpoints of a model alias (configs.alias); returns (sr, enc_ckpt, dec_ckpt).

    When encoder and decoder share an experiment directory (the *_sym aliases) they share one
    state dict, as in the reference's releases where both step counts come from one training run.
    """
    sr, enc_tag, tx_steps, dec_tag, rx_steps = configs.alias(model)
    enc = write_experiment(root, enc_tag, tx_steps, seed)
    dec = write_experiment(root, dec_tag, rx_steps, seed)
    return sr, enc, dec


def synth_audio(seed, stream, length, amp=0.1):
    """x = amp * randn, clipped to [-1, 1] (SURVEY.md section 8d 'synthetic inputs')."""
    x = amp * _randn(seed, ("audio", stream), (length,))
    return np.clip(x, -1.0, 1.0).astype(np.float32)
