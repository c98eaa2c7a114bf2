"""Many logical audio streams multiplexed into one batched codec step.

The reference's ``AudioCodecStreamer`` (bin/stream.py:80-366) serves ONE stream: a PortAudio callback
feeds an encoder thread and a decoder thread through queues, output underruns play silence, and when
the measured latency exceeds ``max_latency`` all queues are cleared and the skipped frames are counted
as drops (bin/stream.py:242-278).  ``BatchedAudioDecStreamer`` keeps those semantics per logical stream
while running every stream of a tick through ONE encode -> RVQ -> pack -> unpack+lookup -> decode pass
on the GPU:

  * real-time streams are isochronous: every tick each stream contributes exactly one frame of
    ``frame_size`` samples; a stream whose frame has not arrived contributes silence and gets an
    underrun counted (the reference plays ``torch.zeros`` on an empty output queue);
  * per-stream reset / join: ``reset_stream(b)`` restores the warmed-up codec state of that stream only;
  * latency / drop accounting per stream with the reference's rule (latency > max_latency: pending
    frames of that stream are discarded and counted);
  * transport between transmitter and receiver is the 80 bit/frame payload of ``wire.py``.
"""
import collections
import time

import numpy as np
import torch

from . import native


class BatchedAudioDecStreamer:
    def __init__(self, audiodec, frame_size, sample_rate=48000, gain=1.0, max_latency=0.1, use_wire_format=True):
        self.tx, self.rx, self.dec = audiodec.tx_encoder, audiodec.rx_encoder, audiodec.decoder
        self.n = self.tx.num_streams
        assert self.dec.num_streams == self.n
        assert frame_size % self.tx.hop == 0, f"frame_size({frame_size}) must be a multiple of codec hop_length({self.tx.hop})!"
        assert frame_size // self.tx.hop <= self.tx.max_frames, "frame_size exceeds max_frames * hop"
        self.frame_size, self.sample_rate, self.gain, self.max_latency = frame_size, sample_rate, gain, max_latency
        self.use_wire = use_wire_format
        self.dev = self.tx._dev()
        self.pending = [collections.deque() for _ in range(self.n)]       # (timestamp, frame) per stream
        self.n_frames = [0] * self.n
        self.underruns = [0] * self.n
        self.frame_drops = [0] * self.n
        self.latencies = [[] for _ in range(self.n)]
        self.encoder_times, self.decoder_times = [], []
        self.payload_bytes = 0
        self._x = torch.zeros(self.n, 1, frame_size, dtype=torch.float32, device=self.dev)
        self._host = torch.zeros(self.n, 1, frame_size, dtype=torch.float32).pin_memory() if torch.cuda.is_available() else None

    # ---- per-stream control ----
    def push(self, stream, frame):
        """Queue one frame (frame_size samples, float32 in [-1, 1]) for `stream`."""
        frame = np.asarray(frame, np.float32).reshape(-1)
        assert frame.shape[0] == self.frame_size
        self.pending[stream].append((time.time(), frame * self.gain))

    def reset_stream(self, stream, warm=True):
        """A stream (re)joins: codec state of that stream back to the warmed-up state, queue flushed."""
        self.pending[stream].clear()
        self.tx.reset_stream(stream, warm)
        self.dec.reset_stream(stream, warm)

    # ---- one tick: every stream advances by one frame ----
    def tick(self):
        """Consume at most one pending frame per stream, run the batch, return (n, frame_size) float32 output."""
        now = time.time()
        stamps = [None] * self.n
        self._host.zero_()
        for s in range(self.n):
            q = self.pending[s]
            # the reference's drop rule, per stream: if the oldest pending frame is already too late, discard the
            # backlog (bin/stream.py:259-266)
            if q and now - q[0][0] > self.max_latency:
                self.frame_drops[s] += len(q)
                q.clear()
            if q:
                stamps[s], frame = q.popleft()
                self._host[s, 0] = torch.from_numpy(frame)
            else:
                self.underruns[s] += 1
            self.n_frames[s] += 1
        self._x.copy_(self._host, non_blocking=True)
        t0 = time.time()
        with torch.no_grad():
            idx = self.tx.quantize(self.tx.encode(self._x))
            if self.use_wire:
                payload = self.tx.pack(idx, check=False)                    # what would cross the network: 10 bytes / frame / stream
                self.payload_bytes += payload.numel()
            torch.cuda.synchronize(self.dev)
            t1 = time.time()
            zq = self.rx.lookup_packed(payload) if self.use_wire else self.rx.lookup(idx)
            y = self.dec.decode(zq)
            out = y[:, 0].cpu().numpy()
        native.raise_on_device_flags("BatchedAudioDecStreamer.tick")     # the copy synchronised: surface device-side failures
        t2 = time.time()
        self.encoder_times.append(t1 - t0)
        self.decoder_times.append(t2 - t1)
        for s in range(self.n):
            if stamps[s] is not None:
                self.latencies[s].append(t2 - stamps[s])
        return out

    def stats(self):
        lat = np.concatenate([np.asarray(l) for l in self.latencies if l]) if any(self.latencies) else np.zeros(1)
        return {
            "streams": self.n, "ticks": max(self.n_frames) if self.n_frames else 0,
            "encoder_ms_mean": float(np.mean(self.encoder_times) * 1e3) if self.encoder_times else 0.0,
            "decoder_ms_mean": float(np.mean(self.decoder_times) * 1e3) if self.decoder_times else 0.0,
            "latency_ms_mean": float(lat.mean() * 1e3), "latency_ms_max": float(lat.max() * 1e3),
            "underruns": int(sum(self.underruns)), "frame_drops": int(sum(self.frame_drops)),
            "payload_kbps_per_stream": (8.0 * self.payload_bytes / max(sum(self.n_frames), 1)) * (self.sample_rate / self.frame_size) / 1e3,
        }

    def print_stats(self):
        st = self.stats()
        print("#" * 80)
        print(f"streams:                           {st['streams']}  ticks: {st['ticks']}")
        print(f"encoder processing time (ms):      {st['encoder_ms_mean']:.2f}")
        print(f"decoder processing time (ms):      {st['decoder_ms_mean']:.2f}")
        print(f"system latency (ms):               {st['latency_ms_mean']:.2f} (max {st['latency_ms_max']:.2f})")
        print(f"underruns / frame drops:           {st['underruns']} / {st['frame_drops']}")
        print(f"payload per stream (kbps):         {st['payload_kbps_per_stream']:.2f}")
        print("#" * 80)
