"""Codec loader base and real-time streamer (mirrors /root/reference/bin/stream.py).

``AudioCodec`` (bin/stream.py:23-77) is the load path: read ``config.yml`` next to the checkpoint,
build the model, warm it up.  ``AudioCodecStreamer`` (bin/stream.py:80-366) is the device-agnostic
queue/thread runtime around ``_encode`` / ``_decode``; it is not on the accelerated path, so it is
kept behaviourally identical (same queues, latency bookkeeping, frame-drop rule, statistics
printout); WAV dumps go through ``scipy.io.wavfile`` because torchaudio is not a dependency here.
"""
import abc
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


def _save_wav(path, audio, sample_rate):
    """audio: (channels, samples) float tensor in [-1, 1] -> 16-bit PCM WAV."""
    from scipy.io import wavfile
    data = (audio.transpose(1, 0).numpy() * 32767.0).round().astype(np.int16)
    wavfile.write(path, int(sample_rate), data)


class AudioCodecStreamer(abc.ABC):
    """Microphone -> encoder thread -> decoder thread -> speaker (bin/stream.py:80-366)."""

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
        self.input_device = input_device
        self.output_device = output_device
        self.input_channels = input_channels
        self.output_channels = output_channels
        self.frame_size = frame_size
        self.sample_rate = sample_rate
        self.gain = gain
        self.max_latency = max_latency
        self.tx_encoder = tx_encoder
        self.tx_device = tx_device
        print(f"Encoder device: {tx_device}")
        self.rx_encoder = rx_encoder
        self.decoder = decoder
        self.rx_device = rx_device
        print(f"Decoder device: {rx_device}")
        self.encoder_queue = queue.Queue()
        self.decoder_queue = queue.Queue()
        self.output_queue = queue.Queue()
        self.input_dump = []
        self.output_dump = []
        self.input_dump_filename = None
        self.output_dump_filename = None
        self.frame_drops = 0
        self.n_frames = 0
        self.encoder_times = []
        self.decoder_times = []
        self.latency_queue = queue.Queue()
        self.latencies = []

    @abc.abstractmethod
    def _encode(self, x):
        pass

    @abc.abstractmethod
    def _decode(self, x):
        pass

    def _run_encoder(self):
        while threading.main_thread().is_alive():
            try:
                x = self.encoder_queue.get(timeout=1)
            except queue.Empty:
                continue
            start = time.time()
            x = x.to(self.tx_device)
            with torch.no_grad():
                if self.tx_encoder is not None:
                    x = self._encode(x)
            self.encoder_times.append(time.time() - start)
            self.decoder_queue.put(x)

    def _run_decoder(self):
        while threading.main_thread().is_alive():
            try:
                x = self.decoder_queue.get(timeout=1)
            except queue.Empty:
                continue
            start = time.time()
            x = x.to(self.rx_device)
            with torch.no_grad():
                if (self.rx_encoder is not None) and (self.decoder is not None):
                    x = self._decode(x)
            self.decoder_times.append(time.time() - start)
            self.output_queue.put(x)

    def _process(self, data):
        data = data * self.gain
        input_data = torch.from_numpy(data).transpose(1, 0).contiguous()  # channels x frame_size
        if self.input_dump_filename is not None:
            self.input_dump.append(input_data)
        input_data = input_data.unsqueeze(0)
        self.encoder_queue.put(input_data)
        self.latency_queue.put(time.time())
        try:
            output_data = self.output_queue.get_nowait()
            latency = time.time() - self.latency_queue.get_nowait()
            self.latencies.append(latency)
            # clear queues if latency gets too high; this leads to frame drops (bin/stream.py:259-266)
            if latency > self.max_latency:
                self.encoder_queue.queue.clear()
                self.decoder_queue.queue.clear()
                self.output_queue.queue.clear()
                while not self.latency_queue.empty():
                    self.frame_drops += 1
                    self.latency_queue.get_nowait()
        except queue.Empty:
            output_data = torch.zeros(1, self.output_channels, self.frame_size)
        output_data = output_data.squeeze(0).detach().cpu()
        self.n_frames += 1
        if self.output_dump_filename is not None:
            self.output_dump.append(output_data)
        return output_data.transpose(1, 0).contiguous().numpy()

    def _callback(self, indata, outdata, frames, _time, status):
        if status:
            print(status)
        outdata[:] = self._process(indata)

    def _exit(self):
        if self.input_dump_filename is not None:
            audio = torch.clamp(torch.cat(self.input_dump, dim=-1), min=-1, max=1)
            _save_wav(self.input_dump_filename, audio, self.sample_rate)
        if self.output_dump_filename is not None:
            audio = torch.clamp(torch.cat(self.output_dump, dim=-1), min=-1, max=1)
            _save_wav(self.output_dump_filename, audio, self.sample_rate)
        with threading.Lock():
            encoder_mean = np.mean(np.array(self.encoder_times) * 1000.0)
            encoder_std = np.std(np.array(self.encoder_times) * 1000.0)
            decoder_mean = np.mean(np.array(self.decoder_times) * 1000.0)
            decoder_std = np.std(np.array(self.decoder_times) * 1000.0)
            latency_mean = np.mean(np.array(self.latencies) * 1000.0)
            latency_std = np.std(np.array(self.latencies) * 1000.0)
        frame_drops_ratio = self.frame_drops / max(self.n_frames, 1)
        print("#" * 80)
        print(f"encoder processing time (ms):      {encoder_mean:.2f} +- {encoder_std:.2f}")
        print(f"decoder processing time (ms):      {decoder_mean:.2f} +- {decoder_std:.2f}")
        print(f"system latency (ms):               {latency_mean:.2f} +- {latency_std:.2f}")
        print(f"frame drops:                       {self.frame_drops} ({frame_drops_ratio * 100:.2f}%)")
        print("#" * 80)

    def enable_filedump(self, input_stream_file: str = None, output_stream_file: str = None):
        if input_stream_file is None and output_stream_file is None:
            raise Exception("At least one of input_stream_file and output_stream_file must be specified.")
        if input_stream_file is not None:
            if not input_stream_file[-4:] == ".wav":
                input_stream_file += ".wav"
            self.input_dump_filename = input_stream_file
        if output_stream_file is not None:
            if not output_stream_file[-4:] == ".wav":
                output_stream_file += ".wav"
            self.output_dump_filename = output_stream_file

    def run(self, latency):
        encoder_thread = threading.Thread(target=self._run_encoder, daemon=True)
        encoder_thread.start()
        decoder_thread = threading.Thread(target=self._run_decoder, daemon=True)
        decoder_thread.start()
        try:
            import sounddevice as sd
            with sd.Stream(
                device=(self.input_device, self.output_device),
                samplerate=self.sample_rate,
                blocksize=self.frame_size,
                dtype=np.float32,
                latency=latency,
                channels=(self.input_channels, self.output_channels),
                callback=self._callback,
            ):
                print("### starting stream [press Return to quit] ###")
                input()
                self._exit()
        except KeyboardInterrupt:
            self._exit()
        except Exception as e:
            print(type(e).__name__ + ": " + str(e))
