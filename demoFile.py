#!/usr/bin/env python3
"""File round trip on the MI355X path: WAV in -> encode -> RVQ codes -> lookup -> decode -> WAV out.

Takes the command line of the reference's demoFile.py (/root/reference/demoFile.py:22-38: --model, -i, -o, --cuda,
--num_threads) so it can stand in for it; WAV I/O goes through scipy.
"""
import argparse
import os

import numpy as np
import torch
from scipy.io import wavfile

from audiodec_amd import native
from audiodec_amd.audiodec import AudioDec, assign_model


def parse_args():
    p = argparse.ArgumentParser(description=__doc__.splitlines()[0])
    p.add_argument("--model", type=str, default="libritts_v1")
    p.add_argument("-i", "--input", type=str, required=True)
    p.add_argument("-o", "--output", type=str, required=True)
    p.add_argument("--cuda", type=int, default=0, help="HIP device index (a negative value selects 'cpu', which this path rejects)")
    p.add_argument("--num_threads", type=int, default=4)
    return p.parse_args()


def read_channels(path, expected_rate):
    """WAV -> (channels, 1, samples) float32 in [-1, 1): every channel becomes one stream."""
    if not os.path.exists(path):
        raise ValueError(f"Input file {path} does not exist!")
    rate, pcm = wavfile.read(path)
    assert rate == expected_rate, f"data ({rate}Hz) is not matched to model ({expected_rate}Hz)!"
    x = pcm.astype(np.float32) / 32768.0 if pcm.dtype == np.int16 else pcm.astype(np.float32)
    x = x[:, None] if x.ndim == 1 else x
    return torch.from_numpy(np.ascontiguousarray(x.T))[:, None, :], rate


def main():
    args = parse_args()
    device = f"cuda:{args.cuda}" if args.cuda >= 0 else "cpu"
    torch.set_num_threads(args.num_threads)
    sample_rate, enc_ckpt, dec_ckpt = assign_model(args.model)
    x, rate = read_channels(args.input, sample_rate)

    print("AudioDec initinalizing!")
    codec = AudioDec(tx_device=device, rx_device=device, num_streams=x.shape[0])
    codec.load_transmitter(enc_ckpt)
    codec.load_receiver(enc_ckpt, dec_ckpt)

    print("Encode/Decode...")
    with torch.no_grad():
        codes = codec.tx_encoder.quantize(codec.tx_encoder.encode(x.to(device)))
        y = codec.decoder.decode(codec.rx_encoder.lookup(codes))[..., :x.shape[-1]]
    pcm = (y[:, 0].clamp(-1, 1).cpu().numpy().T * 32767.0).round().astype(np.int16)     # samples x channels, PCM_16
    native.raise_on_device_flags("demoFile")               # a device-side failure (bad index, overflow, ...) is an error, not audio
    wavfile.write(args.output, rate, pcm)
    print(f"Output {args.output}!")


if __name__ == "__main__":
    main()
