#!/usr/bin/env python3
"""File round-trip demo on the MI355X path -- same flags and flow as the reference's demoFile.py
(/root/reference/demoFile.py:22-69).  WAV I/O uses scipy (the reference uses soundfile)."""
import argparse
import os

import numpy as np
import torch
from scipy.io import wavfile

from audiodec_amd.audiodec import AudioDec, assign_model


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--model", type=str, default="libritts_v1")
    parser.add_argument("-i", "--input", type=str, required=True)
    parser.add_argument("-o", "--output", type=str, required=True)
    parser.add_argument("--cuda", type=int, default=0)
    parser.add_argument("--num_threads", type=int, default=4)
    args = parser.parse_args()

    if args.cuda < 0:
        tx_device = rx_device = "cpu"          # raises NativeError below: there is no CPU path
    else:
        tx_device = rx_device = f"cuda:{args.cuda}"
    torch.set_num_threads(args.num_threads)

    sample_rate, encoder_checkpoint, decoder_checkpoint = assign_model(args.model)

    print("AudioDec initinalizing!")
    audiodec = AudioDec(tx_device=tx_device, rx_device=rx_device)
    audiodec.load_transmitter(encoder_checkpoint)
    audiodec.load_receiver(encoder_checkpoint, decoder_checkpoint)

    with torch.no_grad():
        if not os.path.exists(args.input):
            raise ValueError(f"Input file {args.input} does not exist!")
        fs, data = wavfile.read(args.input)
        if data.dtype == np.int16:
            data = data.astype(np.float32) / 32768.0
        data = np.atleast_2d(data.astype(np.float32).T).T            # (T, C)
        assert fs == sample_rate, f"data ({fs}Hz) is not matched to model ({sample_rate}Hz)!"
        x = np.expand_dims(data.transpose(1, 0), axis=1)              # (T, C) -> (C, 1, T)
        x = torch.tensor(x, dtype=torch.float).to(tx_device)
        audiodec.tx_encoder.configure(x.shape[0], audiodec.max_frames)
        audiodec.decoder.configure(x.shape[0], audiodec.max_frames)
        print("Encode/Decode...")
        z = audiodec.tx_encoder.encode(x)
        idx = audiodec.tx_encoder.quantize(z)
        zq = audiodec.rx_encoder.lookup(idx)
        y = audiodec.decoder.decode(zq)[:, :, :x.size(-1)]
        y = y.squeeze(1).transpose(1, 0).cpu().numpy()                # T x C
        wavfile.write(args.output, fs, (np.clip(y, -1, 1) * 32767.0).round().astype(np.int16))   # PCM_16
        print(f"Output {args.output}!")


if __name__ == "__main__":
    main()
