"""Codec loader base and live single-stream streamer for the HIP path.

Public surface = what the reference's demos touch (/root/reference/bin/stream.py): ``AudioCodec``
(``load_transmitter`` / ``load_receiver``, :23-77) and ``AudioCodecStreamer`` (constructor keywords,
``enable_filedump``, ``run``, the ``_encode`` / ``_decode`` hooks, :80-366).  The runtime behind it is organised
differently from the reference: a generic ``_Stage`` worker thread per model half, a ``_LatencyLedger`` that owns the
time stamps and the frame-drop rule, and ``_WavSink`` objects for the optional dumps.  Behaviour kept: input gain,
silence when no decoded block is ready, and the rule that a block whose end-to-end latency exceeds ``max_latency``
flushes everything in flight and counts the flushed blocks as drops (bin/stream.py:259-266).  None of this is on the
accelerated path; WAV dumps use ``scipy.io.wavfile``.
"""
import abc
import collections
import os
import queue
import threading
import time
from typing import Union

import numpy as np
import torch
import yaml


class AudioCodec(abc.ABC):
    def __init__(self, tx_device: str = "cpu", rx_device: str = "cpu", receptive_length: int = 8192):
        self.tx_device = tx_device
        self.rx_device = rx_device
        self.receptive_length = receptive_length
        self.tx_encoder = None
        self.rx_encoder = None
        self.decoder = None

    @abc.abstractmethod
    def _load_encoder(self, checkpoint):
        pass

    @abc.abstractmethod
    def _load_decoder(self, checkpoint):
        pass

    def _load_config(self, checkpoint, config_name="config.yml"):
        dirname = os.path.dirname(checkpoint)
        config_path = os.path.join(dirname, config_name)
        with open(config_path) as f:
            config = yaml.load(f, Loader=yaml.Loader)
        return config

    def load_transmitter(self, encoder_checkpoint):
        # bin/stream.py:56-62
        assert os.path.exists(encoder_checkpoint), f"{encoder_checkpoint} does not exist!"
        self.tx_encoder = self._load_encoder(encoder_checkpoint)
        self.tx_encoder.eval().to(self.tx_device)
        self.tx_encoder.initial_encoder(self.receptive_length, self.tx_device)
        print("Load tx_encoder: %s" % (encoder_checkpoint))

    def load_receiver(self, encoder_checkpoint, decoder_checkpoint):
        # bin/stream.py:65-77
        assert os.path.exists(encoder_checkpoint), f"{encoder_checkpoint} does not exist!"
        self.rx_encoder = self._load_encoder(encoder_checkpoint)
        self.rx_encoder.eval().to(self.rx_device)
        zq = self.rx_encoder.initial_encoder(self.receptive_length, self.rx_device)
        print("Load rx_encoder: %s" % (encoder_checkpoint))

        assert os.path.exists(decoder_checkpoint), f"{decoder_checkpoint} does not exist!"
        self.decoder = self._load_decoder(decoder_checkpoint)
        self.decoder.eval().to(self.rx_device)
        self.decoder.initial_decoder(zq)
        print("Load decoder: %s" % (decoder_checkpoint))


class _WavSink:
    """Collects (channels, samples) blocks and writes one clipped 16-bit PCM file when closed."""

    def __init__(self, path, sample_rate):
        self.path = path if path.endswith(".wav") else path + ".wav"
        self.sample_rate = int(sample_rate)
        self.blocks = []

    def push(self, block):
        self.blocks.append(block)

    def close(self):
        if not self.blocks:
            return
        from scipy.io import wavfile
        audio = torch.cat(self.blocks, dim=-1).clamp(-1.0, 1.0)
        pcm = (audio.transpose(1, 0).numpy() * 32767.0).round().astype(np.int16)
        wavfile.write(self.path, self.sample_rate, pcm)


class _LatencyLedger:
    """Time stamps of the blocks in flight, completed latencies, and the drop count."""

    def __init__(self, limit_s):
        self.limit_s = limit_s
        self.in_flight = collections.deque()
        self.done = []
        self.drops = 0
        self.blocks = 0

    def submitted(self):
        self.in_flight.append(time.time())

    def completed(self):
        """Latency of the oldest block in flight; True when it breaks the limit."""
        lat = time.time() - self.in_flight.popleft()
        self.done.append(lat)
        return lat > self.limit_s

    def flush(self):
        self.drops += len(self.in_flight)
        self.in_flight.clear()


class _Stage(threading.Thread):
    """One model half as a daemon worker: inbox -> fn(x on device) -> outbox, with per-block wall times."""

    def __init__(self, name, fn, device, inbox, outbox):
        super().__init__(name=name, daemon=True)
        self.fn, self.device, self.inbox, self.outbox = fn, device, inbox, outbox
        self.times = []

    def run(self):
        while threading.main_thread().is_alive():
            try:
                x = self.inbox.get(timeout=1)
            except queue.Empty:
                continue
            t0 = time.time()
            with torch.no_grad():
                y = self.fn(x.to(self.device))
            self.times.append(time.time() - t0)
            self.outbox.put(y)


def _ms(values):
    a = np.asarray(values, dtype=np.float64) * 1e3
    return (float(a.mean()), float(a.std())) if a.size else (float("nan"), float("nan"))


class AudioCodecStreamer(abc.ABC):
    """Sound card -> encoder stage -> decoder stage -> sound card, one block of ``frame_size`` samples per callback."""

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
        tx_encoder=None,
        tx_device: str = "cpu",
        rx_encoder=None,
        decoder=None,
        rx_device: str = "cpu",
    ):
        self.input_device, self.output_device = input_device, output_device
        self.input_channels, self.output_channels = input_channels, output_channels
        self.frame_size, self.sample_rate = frame_size, sample_rate
        self.gain, self.max_latency = gain, max_latency
        self.tx_encoder, self.tx_device = tx_encoder, tx_device
        self.rx_encoder, self.decoder, self.rx_device = rx_encoder, decoder, rx_device
        print(f"Encoder device: {tx_device}")
        print(f"Decoder device: {rx_device}")
        self._to_tx, self._to_rx, self._to_out = queue.Queue(), queue.Queue(), queue.Queue()
        self._ledger = _LatencyLedger(max_latency)
        self._sinks = {"in": None, "out": None}
        have_tx = tx_encoder is not None
        have_rx = rx_encoder is not None and decoder is not None
        self._tx = _Stage("adk-tx", self._encode if have_tx else (lambda x: x), tx_device, self._to_tx, self._to_rx)
        self._rx = _Stage("adk-rx", self._decode if have_rx else (lambda x: x), rx_device, self._to_rx, self._to_out)

    # ---- model hooks supplied by the subclass (utils/audiodec.py:100-106) ----
    @abc.abstractmethod
    def _encode(self, x):
        pass

    @abc.abstractmethod
    def _decode(self, x):
        pass

    # ---- statistics the callers may read ----
    @property
    def frame_drops(self):
        return self._ledger.drops

    @property
    def n_frames(self):
        return self._ledger.blocks

    def tick(self, block):
        """Body of the audio callback: block (frame_size, in_channels) float32 in, (frame_size, out_channels) out."""
        x = torch.from_numpy(block * self.gain).transpose(1, 0).contiguous()          # channels x frame_size
        if self._sinks["in"] is not None:
            self._sinks["in"].push(x)
        self._to_tx.put(x.unsqueeze(0))
        self._ledger.submitted()
        try:
            y = self._to_out.get_nowait()
        except queue.Empty:
            y = torch.zeros(1, self.output_channels, self.frame_size)                  # nothing decoded yet: silence
        else:
            if self._ledger.completed():                                               # too late: start over
                for q in (self._to_tx, self._to_rx, self._to_out):
                    with q.mutex:
                        q.queue.clear()
                self._ledger.flush()
        y = y.squeeze(0).detach().cpu()
        if self._ledger.blocks % 16 == 0:                          # the copy above synchronised: a cheap place to surface
            self._check_device()                                   # device-side failures as exceptions (native.py)
        self._ledger.blocks += 1
        if self._sinks["out"] is not None:
            self._sinks["out"].push(y)
        return y.transpose(1, 0).contiguous().numpy()

    def _check_device(self):
        if str(self.tx_device).startswith("cuda") or str(self.rx_device).startswith("cuda"):
            from . import native
            native.raise_on_device_flags("AudioCodecStreamer")

    def enable_filedump(self, input_stream_file: str = None, output_stream_file: str = None):
        if input_stream_file is None and output_stream_file is None:
            raise Exception("At least one of input_stream_file and output_stream_file must be specified.")
        if input_stream_file is not None:
            self._sinks["in"] = _WavSink(input_stream_file, self.sample_rate)
        if output_stream_file is not None:
            self._sinks["out"] = _WavSink(output_stream_file, self.sample_rate)

    def report(self):
        """Write the dumps and print the run statistics."""
        for sink in self._sinks.values():
            if sink is not None:
                sink.close()
        rows = (("encoder processing time (ms)", _ms(self._tx.times)), ("decoder processing time (ms)", _ms(self._rx.times)),
                ("system latency (ms)", _ms(self._ledger.done)))
        bar = "#" * 80
        print(bar)
        for label, (mean, std) in rows:
            print(f"{label + ':':35s}{mean:.2f} +- {std:.2f}")
        share = 100.0 * self._ledger.drops / max(self._ledger.blocks, 1)
        print(f"{'frame drops:':35s}{self._ledger.drops} ({share:.2f}%)")
        print(bar)

    def run(self, latency):
        """Open the duplex sound-card stream and pump blocks until Return is pressed (needs ``sounddevice``)."""
        self._tx.start()
        self._rx.start()

        def callback(indata, outdata, frames, _time, status):
            if status:
                print(status)
            outdata[:] = self.tick(indata)

        try:
            import sounddevice as sd
            stream = sd.Stream(device=(self.input_device, self.output_device), samplerate=self.sample_rate,
                               blocksize=self.frame_size, dtype=np.float32, latency=latency,
                               channels=(self.input_channels, self.output_channels), callback=callback)
            with stream:
                print("### starting stream [press Return to quit] ###")
                input()
            self.report()
        except KeyboardInterrupt:
            self.report()
        except Exception as e:                                   # the reference prints and swallows (bin/stream.py:365-366)
            print(type(e).__name__ + ": " + str(e))
