#!/usr/bin/env python3
"""Live mic -> speaker loop on the MI355X path -- same flags and flow as the reference's
demoStream.py (/root/reference/demoStream.py:19-79).  Needs `sounddevice` at run time."""
import argparse

import torch

from audiodec_amd.audiodec import AudioDec, AudioDecStreamer, assign_model


def main():
    parser = argparse.ArgumentParser()
    parser.add_argument("--model", type=str, default="libritts_sym")
    parser.add_argument("-i", "--input", type=str, default="input.wav")
    parser.add_argument("-o", "--output", type=str, default="output.wav")
    parser.add_argument("--tx_cuda", type=int, default=0)   # reference default -1 (cpu); there is no CPU path here
    parser.add_argument("--rx_cuda", type=int, default=0)
    parser.add_argument("--input_device", type=int, default=1)
    parser.add_argument("--output_device", type=int, default=4)
    parser.add_argument("--frame_size", type=int, default=1500)
    parser.add_argument("--num_threads", type=int, default=4)
    args = parser.parse_args()

    tx_device = "cpu" if args.tx_cuda < 0 else f"cuda:{args.tx_cuda}"
    rx_device = "cpu" if args.rx_cuda < 0 else f"cuda:{args.rx_cuda}"
    torch.set_num_threads(args.num_threads)

    sample_rate, encoder_checkpoint, decoder_checkpoint = assign_model(args.model)

    print("AudioDec initinalizing!")
    audiodec = AudioDec(tx_device=tx_device, rx_device=rx_device)
    hop = audiodec.get_hop_length(encoder_checkpoint)
    assert args.frame_size % hop == 0, f"frame_size {args.frame_size} must be a multiple of the hop {hop}"
    audiodec.max_frames = max(audiodec.max_frames, args.frame_size // hop)
    audiodec.load_transmitter(encoder_checkpoint)
    audiodec.load_receiver(encoder_checkpoint, decoder_checkpoint)

    print("Streamer initinalizing!")
    streamer = AudioDecStreamer(
        input_device=args.input_device, output_device=args.output_device, frame_size=args.frame_size,
        sample_rate=sample_rate, tx_encoder=audiodec.tx_encoder, tx_device=tx_device,
        rx_encoder=audiodec.rx_encoder, decoder=audiodec.decoder, rx_device=rx_device)
    streamer.enable_filedump(input_stream_file=args.input, output_stream_file=args.output)

    print("Ready to run!")
    latency = "low"
    streamer.run(latency)


if __name__ == "__main__":
    main()
