"""CPU ORACLE for the bit-packed code wire format  --  TEST INFRASTRUCTURE, NOT PRODUCT CODE.

The reference never serialises codes (it hands the int64 index tensor from the encoder thread to
the decoder thread through a queue.Queue, bin/stream.py:224,230); the wire format is this
repository's "next" row, so the oracle here is the numpy restatement of the
format specified in include/audiodec_hip.h: per frame, code q = idx[q] - size*q
(layers/vq_module.py:145-146 adds that offset) in bits [q*bits, (q+1)*bits), LSB-first.
Parity pin: tests/golden/wire.npz holds a known-answer payload written out by hand-checkable
arithmetic (make_wire_golden below); the GPU kernels must match it and this oracle bit for bit.
"""
import numpy as np


def frame_bytes(n_q, bits):
    return (n_q * bits + 7) // 8


def pack(idx, bits, size):
    """idx (n_q, n_rows) int64 -> payload (n_rows, frame_bytes) uint8."""
    n_q, n_rows = idx.shape
    fb = frame_bytes(n_q, bits)
    out = np.zeros((n_rows, fb), np.uint8)
    for row in range(n_rows):
        acc = 0
        for q in range(n_q):
            code = int(idx[q, row]) - size * q
            assert 0 <= code < size
            acc |= code << (q * bits)
        out[row] = np.frombuffer(acc.to_bytes(fb, "little"), np.uint8)
    return out


def unpack(payload, n_q, bits, size):
    n_rows, fb = payload.shape
    idx = np.zeros((n_q, n_rows), np.int64)
    for row in range(n_rows):
        acc = int.from_bytes(payload[row].tobytes(), "little")
        for q in range(n_q):
            idx[q, row] = ((acc >> (q * bits)) & ((1 << bits) - 1)) + size * q
    return idx


def make_wire_golden():
    """Known answers checkable by hand."""
    # 8 codes of 10 bit: code q = q+1 -> little-endian integer sum (q+1) << 10q
    idx = (np.arange(8)[:, None] + 1 + 1024 * np.arange(8)[:, None]).astype(np.int64)
    acc = sum((q + 1) << (10 * q) for q in range(8))
    expect = np.frombuffer(acc.to_bytes(10, "little"), np.uint8)[None, :]
    assert np.array_equal(pack(idx, 10, 1024), expect)
    # all-ones codes -> all-ones payload
    idx1 = (1023 + 1024 * np.arange(8)[:, None]).astype(np.int64)
    assert np.array_equal(pack(idx1, 10, 1024), np.full((1, 10), 255, np.uint8))
    return idx, expect
