"""Drop-in for /root/reference/utils/audiodec.py: ``AudioDec``, ``AudioDecStreamer``, ``assign_model``.

Same names, same arguments, same errors; the model objects behind ``tx_encoder`` /
``rx_encoder`` / ``decoder`` are the HIP-backed stream generators of stream_generator.py.
Extra (defaulted) keywords: ``num_streams`` (independent streams per object; the reference is
batch-1 only) and ``max_frames`` (largest chunk, in hops, handled by one kernel sequence).
"""
import math
import os
from typing import Union

import torch

from . import lazy_guard, native
from .configs import assign_model  # noqa: F401  (utils/audiodec.py:109-179)
from .stream import AudioCodec, AudioCodecStreamer
from .stream_generator import AutoEncoderStreamGenerator as generator_audiodec
from .stream_generator import HiFiGANStreamGenerator as generator_hifigan


class AudioDec(AudioCodec):
    def __init__(
        self,
        tx_device: str = "cpu",
        rx_device: str = "cpu",
        receptive_length: int = 8192,  # actual number is 7209 for symAD_vctk_48000_hop300
        num_streams: int = 1,
        max_frames: int = 16,
        guard: bool = None,
    ):
        # The signature keeps the reference's 'cpu' defaults (utils/audiodec.py:20-30) so that callers port unchanged, but
        # this package has no CPU compute path: 'cpu' becomes the first HIP device (with a warning), and without a HIP
        # device the constructor fails here, not at the first kernel launch.
        tx_device, rx_device = native.resolve_device(tx_device), native.resolve_device(rx_device)
        super(AudioDec, self).__init__(tx_device=tx_device, rx_device=rx_device, receptive_length=receptive_length)
        for d in (tx_device, rx_device):
            native.require_gpu(d)
        self.num_streams = num_streams
        self.max_frames = max_frames
        # guard=True: every program step is checked on the device and a split-f16 range overflow is repaired in place by the
        # exact-f32 kernels (stream_generator.set_guard); False for callers that keep several steps in flight (bench.py);
        # None (default): the generators' own default -- on, unless the environment says ADK_GUARD=0
        self.guard = guard

    def _load_encoder(self, checkpoint):
        # utils/audiodec.py:32-42
        config = self._load_config(checkpoint)
        if config["model_type"] in ["symAudioDec", "symAudioDecUniv"]:
            encoder = generator_audiodec
        else:
            raise NotImplementedError(f"Encoder type {config['model_type']} is not supported!")
        encoder = encoder(**config["generator_params"])
        encoder.load_state_dict(torch.load(checkpoint, map_location="cpu")["model"]["generator"])
        return encoder.configure(self.num_streams, self.max_frames).set_guard(self.guard)

    def _load_decoder(self, checkpoint):
        # utils/audiodec.py:44-56
        config = self._load_config(checkpoint)
        if config["model_type"] in ["symAudioDec", "symAudioDecUniv"]:
            decoder = generator_audiodec
        elif config["model_type"] in ["HiFiGAN", "UnivNet"]:
            decoder = generator_hifigan
        else:
            raise NotImplementedError(f"Decoder {config['model_type']} is not supported!")
        decoder = decoder(**config["generator_params"])
        decoder.load_state_dict(torch.load(checkpoint, map_location="cpu")["model"]["generator"])
        return decoder.configure(self.num_streams, self.max_frames).set_guard(self.guard)

    def load_transmitter(self, encoder_checkpoint):
        super().load_transmitter(encoder_checkpoint)
        self._share_logs()

    def _share_logs(self):
        """One lazy_guard.CallLog per device for the generators of this object: results that feed each other (encode -> quantize -> lookup ->
        decode) are verified -- and, after an f16 range overflow, repeated -- in call order."""
        by_dev = {}
        for g in (self.tx_encoder, self.rx_encoder, self.decoder):
            if g is None or g._device is None:
                continue
            key = str(g._device)
            if key not in by_dev:
                by_dev[key] = g._log if g._log is not None else lazy_guard.CallLog(g._dev())
            g.share_log(by_dev[key])

    def settle(self):
        """Every direct call made through this object's generators is verified (and repaired if need be) when this returns."""
        for g in (self.tx_encoder, self.rx_encoder, self.decoder):
            if g is not None:
                g.settle()

    def load_receiver(self, encoder_checkpoint, decoder_checkpoint):
        # bin/stream.py:65-77.  The receiver-side encoder only supplies the codebook and the warm-up
        # zq (its conv state is never stepped again), so it carries a single stream.
        assert os.path.exists(encoder_checkpoint), f"{encoder_checkpoint} does not exist!"
        self.rx_encoder = self._load_encoder(encoder_checkpoint).configure(1, self.max_frames)
        self.rx_encoder.eval().to(self.rx_device)
        zq = self.rx_encoder.initial_encoder(self.receptive_length, self.rx_device)
        print("Load rx_encoder: %s" % (encoder_checkpoint))

        assert os.path.exists(decoder_checkpoint), f"{decoder_checkpoint} does not exist!"
        self.decoder = self._load_decoder(decoder_checkpoint)
        self.decoder.eval().to(self.rx_device)
        self.decoder.initial_decoder(zq)
        print("Load decoder: %s" % (decoder_checkpoint))
        self._share_logs()

    def get_hop_length(self, checkpoint):
        # utils/audiodec.py:58-62
        assert os.path.exists(checkpoint), f"{checkpoint} does not exist!"
        config = self._load_config(checkpoint)
        return math.prod(config["generator_params"]["enc_strides"])


class AudioDecStreamer(AudioCodecStreamer):
    def __init__(
        self,
        input_device: Union[str, int],
        output_device: Union[str, int],
        input_channels: int = 1,
        output_channels: int = 1,
        frame_size: int = 512,
        sample_rate: int = 48000,
        gain: int = 1.0,
        max_latency: float = 0.1,
        # encoder params
        tx_encoder=None,
        tx_device: str = "cpu",
        # decoder params
        rx_encoder=None,
        decoder=None,
        rx_device: str = "cpu",
    ):
        tx_device, rx_device = native.resolve_device(tx_device), native.resolve_device(rx_device)
        super(AudioDecStreamer, self).__init__(
            input_device=input_device, output_device=output_device, input_channels=input_channels,
            output_channels=output_channels, frame_size=frame_size, sample_rate=sample_rate, gain=gain,
            max_latency=max_latency, tx_encoder=tx_encoder, tx_device=tx_device, rx_encoder=rx_encoder,
            decoder=decoder, rx_device=rx_device)

    def _encode(self, x):
        x = self.tx_encoder.encode(x)          # utils/audiodec.py:100-102
        return self.tx_encoder.quantize(x)

    def _decode(self, x):
        x = self.rx_encoder.lookup(x)          # utils/audiodec.py:104-106
        return self.decoder.decode(x)
