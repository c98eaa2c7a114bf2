"""Bit-packed code payloads between transmitter and receiver.

The reference hands the int64 index tensor from the encoder thread to the decoder thread through a
``queue.Queue`` (bin/stream.py:224,230) and never serialises it.  Here ``pack_codes`` turns the
emitted indices into the 80 bit/frame payload the codec's 12.8 kbps figure implies (README.md:6),
``unpack_codes`` inverts it, and ``lookup_packed`` decodes a payload straight to ``zq``
(unpack fused into ResidualVQ.lookup, layers/vq_module.py:159-161).  All three run as HIP kernels
(``audiodec_amd/csrc/wire.hip``) through the C ABI.
"""
import ctypes as C

import torch

from . import native


def code_bits(codebook_size):
    return max(1, (int(codebook_size) - 1).bit_length())


def frame_bytes(n_q, codebook_size):
    return (n_q * code_bits(codebook_size) + 7) // 8


def pack_codes(idx, codebook_size=1024, check=True):
    """idx (n_q, T) or (n_q, B, T) int64 on a HIP device -> uint8 payload (B, T, frame_bytes).

    check=True (the public default): the call SYNCHRONISES the device and raises on any pending device flag -- an index
    that is not a code of its stage (ValueError; it would otherwise travel as code 0), or anything an earlier asynchronous
    launch reported -- so a payload this function returns is safe to ship.  check=False is for real-time tick paths
    that keep the device running (batched_streamer passes it): a bad index is then packed as code 0 and only recorded in
    the sticky device flags, which the caller's next synchronisation point turns into the exception
    (native.raise_on_device_flags: streamer ticks, demoFile, the offline drivers)."""
    dev = native.require_gpu(idx.device)
    if idx.dim() == 2:
        idx = idx.unsqueeze(1)
    n_q, B, T = idx.shape
    idx = idx.to(torch.int64).contiguous()
    bits = code_bits(codebook_size)
    out = torch.empty(B, T, (n_q * bits + 7) // 8, dtype=torch.uint8, device=dev)
    native.check(native.lib().adk_codes_pack(C.c_void_p(idx.data_ptr()), C.c_void_p(out.data_ptr()), B * T, n_q, bits,
                                             int(codebook_size), native.current_stream(dev)), "adk_codes_pack")
    if check:
        native.raise_on_device_flags("pack_codes")
    return out


def unpack_codes(payload, n_q, codebook_size=1024):
    """payload (B, T, frame_bytes) uint8 -> idx (n_q, B, T) int64 (squeezed to (n_q, T) for B == 1)."""
    dev = native.require_gpu(payload.device)
    B, T, fb = payload.shape
    bits = code_bits(codebook_size)
    if fb != (n_q * bits + 7) // 8:
        raise ValueError(f"payload frames are {fb} bytes, expected {(n_q * bits + 7) // 8}")
    payload = payload.contiguous()
    idx = torch.empty(n_q, B * T, dtype=torch.int64, device=dev)
    native.check(native.lib().adk_codes_unpack(C.c_void_p(payload.data_ptr()), C.c_void_p(idx.data_ptr()), B * T, n_q, bits,
                                               int(codebook_size), native.current_stream(dev)), "adk_codes_unpack")
    idx = idx.reshape(n_q, B, T)
    return idx.squeeze(1) if B == 1 else idx


def lookup_packed(payload, codebook, n_q, codebook_size=1024):
    """payload (B, T, frame_bytes) + stacked codebook (n_q*size, dim) -> zq (B, T, dim)."""
    dev = native.require_gpu(payload.device)
    B, T, fb = payload.shape
    bits = code_bits(codebook_size)
    if fb != (n_q * bits + 7) // 8:
        raise ValueError(f"payload frames are {fb} bytes, expected {(n_q * bits + 7) // 8}")
    payload = payload.contiguous()
    dim = codebook.shape[1]
    zq = torch.empty(B, T, dim, dtype=torch.float32, device=dev)
    native.check(native.lib().adk_codes_lookup(C.c_void_p(payload.data_ptr()), C.c_void_p(codebook.data_ptr()),
                                               C.c_void_p(zq.data_ptr()), B * T, n_q, bits, int(codebook_size), dim,
                                               native.current_stream(dev)), "adk_codes_lookup")
    return zq
