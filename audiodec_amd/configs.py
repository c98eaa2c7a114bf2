"""Model table for the AudioDec streaming path.

The reference reads shapes from ``exp/<tag>/config.yml`` (``generator_params`` block,
/root/reference/bin/stream.py:48-53) and maps model aliases to checkpoint paths in
``assign_model`` (/root/reference/utils/audiodec.py:109-179).  This module restates that data:
the alias table (same names, same relative paths, same NotImplementedError) and, for synthetic
checkpoints / tests / bench, the ``generator_params`` of every experiment tag the aliases touch.
The loader itself (checkpoint.py) still reads whatever ``config.yml`` sits next to the ``.pkl``.
"""
import os
import copy

# --------------------------------------------------------------------------------------------
# generator_params per experiment tag (values as in the reference's exp/*/config.yml dumps)
# --------------------------------------------------------------------------------------------

def _ae(strides=(3, 4, 5, 5), codebook_num=8, codec="audiodec", use_weight_norm=None):
    p = dict(
        input_channels=1, output_channels=1, encode_channels=32, decode_channels=32,
        code_dim=64, codebook_num=codebook_num, codebook_size=1024, bias=True,
        enc_ratios=[2, 4, 8, 16], dec_ratios=[16, 8, 4, 2],
        enc_strides=list(strides), dec_strides=list(reversed(strides)),
        mode="causal", codec=codec, projector="conv1d", quantier="residual_vq",
    )
    if use_weight_norm is not None:
        p["use_weight_norm"] = use_weight_norm
    return p


def _voc(kernel_sizes, groups, stats, dilations=None, use_additional_convs=True):
    n = len(kernel_sizes)
    return dict(
        in_channels=64, out_channels=1, channels=512, kernel_size=7,
        upsample_scales=[5, 5, 4, 3], upsample_kernel_sizes=[10, 10, 8, 6],
        resblock_kernel_sizes=list(kernel_sizes),
        resblock_dilations=[[1, 3, 5] for _ in range(n)] if dilations is None else dilations,
        groups=groups, bias=True, use_additional_convs=use_additional_convs,
        nonlinear_activation="LeakyReLU", nonlinear_activation_params={"negative_slope": 0.1},
        use_weight_norm=True, stats=stats,
    )


EXPERIMENTS = {
    # tag: (model_type, sampling_rate, generator_params)
    "autoencoder/symAD_vctk_48000_hop300": ("symAudioDec", 48000, _ae()),
    "autoencoder/symAD_libritts_24000_hop300": ("symAudioDec", 24000, _ae()),
    "autoencoder/symADuniv_vctk_48000_hop300": ("symAudioDecUniv", 48000, _ae()),
    "autoencoder/symAAD_vctk_48000_hop300": ("symAudioDec", 48000,
                                             _ae(codec="activate_audiodec", use_weight_norm=True)),
    "autoencoder/symAD_c16_vctk_48000_hop320": ("symAudioDec", 48000,
                                                _ae(strides=(2, 4, 5, 8), codebook_num=16)),
    "denoise/symAD_vctk_48000_hop300": ("symAudioDec", 48000, _ae()),
    "vocoder/AudioDec_v0_symAD_vctk_48000_hop300_clean": (
        "HiFiGAN", 48000, _voc([3, 7, 11], 1, "stats/symAD_vctk_48000_hop300_clean.npy")),
    "vocoder/AudioDec_v1_symAD_vctk_48000_hop300_clean": (
        "HiFiGAN", 48000, _voc([11], 3, "stats/symAD_vctk_48000_hop300_clean.npy")),
    "vocoder/AudioDec_v1_symAD_libritts_24000_hop300_clean": (
        "HiFiGAN", 24000, _voc([11], 3, "stats/symAD_libritts_24000_hop300_clean.npy")),
    "vocoder/AudioDec_v2_symAD_vctk_48000_hop300_clean": (
        "HiFiGAN", 48000, _voc([3], 3, "stats/symAD_vctk_48000_hop300_clean.npy")),
    "vocoder/AudioDec_v3_symADuniv_vctk_48000_hop300_clean": (
        "UnivNet", 48000, _voc([11], 3, "stats/symADuniv_vctk_48000_hop300_clean.npy")),
    # NOT in the reference's exp/: the generator option no released config switches off (HiFiGANResidualBlock
    # use_additional_convs=False, modules/residual_block.py:93-106: x + convs1(act(x)) without the second conv), as v1- and
    # v0-shaped vocoders, so that the lowering of that branch has reference fixtures too (EXTRA_ALIASES below)
    "autoencoder/test_stereo_symAD_vctk_48000_hop300": ("symAudioDec", 48000, dict(_ae(), input_channels=2, output_channels=2)),
    "vocoder/test_v1_noaddl_symAD_vctk_48000_hop300": (
        "HiFiGAN", 48000, _voc([11], 3, "stats/symAD_vctk_48000_hop300_clean.npy", use_additional_convs=False)),
    "vocoder/test_v0_noaddl_symAD_vctk_48000_hop300": (
        "HiFiGAN", 48000, _voc([3, 7, 11], 1, "stats/symAD_vctk_48000_hop300_clean.npy", use_additional_convs=False)),
}


def experiment(tag):
    model_type, sr, params = EXPERIMENTS[tag]
    return model_type, sr, copy.deepcopy(params)


# --------------------------------------------------------------------------------------------
# alias table: same names / paths / error as /root/reference/utils/audiodec.py:109-179
# --------------------------------------------------------------------------------------------
_ALIASES = {
    # name: (sample_rate, encoder tag, tx_steps, decoder tag, rx_steps)
    "libritts_v1": (24000, "autoencoder/symAD_libritts_24000_hop300", 500000,
                    "vocoder/AudioDec_v1_symAD_libritts_24000_hop300_clean", 500000),
    "libritts_sym": (24000, "autoencoder/symAD_libritts_24000_hop300", 500000,
                     "autoencoder/symAD_libritts_24000_hop300", 1000000),
    "vctk_v1": (48000, "autoencoder/symAD_vctk_48000_hop300", 200000,
                "vocoder/AudioDec_v1_symAD_vctk_48000_hop300_clean", 500000),
    "vctk_sym": (48000, "autoencoder/symAD_vctk_48000_hop300", 200000,
                 "autoencoder/symAD_vctk_48000_hop300", 700000),
    "vctk_v0": (48000, "autoencoder/symAD_vctk_48000_hop300", 200000,
                "vocoder/AudioDec_v0_symAD_vctk_48000_hop300_clean", 500000),
    "vctk_v2": (48000, "autoencoder/symAD_vctk_48000_hop300", 200000,
                "vocoder/AudioDec_v2_symAD_vctk_48000_hop300_clean", 500000),
    "vctk_denoise": (48000, "denoise/symAD_vctk_48000_hop300", 200000,
                     "vocoder/AudioDec_v1_symAD_vctk_48000_hop300_clean", 500000),
    "vctk_univ": (48000, "autoencoder/symADuniv_vctk_48000_hop300", 500000,
                  "vocoder/AudioDec_v3_symADuniv_vctk_48000_hop300_clean", 500000),
    "vctk_univ_sym": (48000, "autoencoder/symADuniv_vctk_48000_hop300", 500000,
                      "autoencoder/symADuniv_vctk_48000_hop300", 1000000),
    "vctk_activate_sym": (48000, "autoencoder/symAAD_vctk_48000_hop300", 200000,
                          "autoencoder/symAAD_vctk_48000_hop300", 700000),
    "vctk_c16h320_sym": (48000, "autoencoder/symAD_c16_vctk_48000_hop320", 500000,
                         "autoencoder/symAD_c16_vctk_48000_hop320", 1000000),
}


# Model names that are NOT the reference's (its assign_model raises for them, and so does ours): test models for generator options
# no released alias exercises.  configs.alias / checkpoint_paths / synth.write_model know them; assign_model does not.
EXTRA_ALIASES = {
    # a stereo codec (input_channels = output_channels = 2: AudioDec.py:229-231 folds any other channel count into the batch);
    # no released checkpoint is stereo, the generator takes the parameters all the same
    "test_stereo_sym": (48000, "autoencoder/test_stereo_symAD_vctk_48000_hop300", 200000,
                        "autoencoder/test_stereo_symAD_vctk_48000_hop300", 700000),
    "test_v1_noaddl": (48000, "autoencoder/symAD_vctk_48000_hop300", 200000,
                       "vocoder/test_v1_noaddl_symAD_vctk_48000_hop300", 500000),
    "test_v0_noaddl": (48000, "autoencoder/symAD_vctk_48000_hop300", 200000,
                       "vocoder/test_v0_noaddl_symAD_vctk_48000_hop300", 500000),
}


def alias(model):
    if model in _ALIASES:
        return _ALIASES[model]
    if model in EXTRA_ALIASES:
        return EXTRA_ALIASES[model]
    raise NotImplementedError(f"Model {model} is not supported!")


def checkpoint_paths(model):
    """(sample_rate, encoder_checkpoint, decoder_checkpoint) with cwd-relative 'exp/...' paths, for the reference's aliases and
    for EXTRA_ALIASES."""
    sample_rate, enc_tag, tx_steps, dec_tag, rx_steps = alias(model)
    encoder_checkpoint = os.path.join("exp", *enc_tag.split("/"), f"checkpoint-{tx_steps}steps.pkl")
    decoder_checkpoint = os.path.join("exp", *dec_tag.split("/"), f"checkpoint-{rx_steps}steps.pkl")
    return sample_rate, encoder_checkpoint, decoder_checkpoint


def assign_model(model):
    """utils/audiodec.py:109-179: the reference's 11 names, NotImplementedError for anything else."""
    if model not in _ALIASES:
        raise NotImplementedError(f"Model {model} is not supported!")
    return checkpoint_paths(model)
