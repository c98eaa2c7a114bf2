"""CPU: the wire-format oracle against its hand-checkable known answers and round-trip properties."""
import numpy as np

from oracle import wire_oracle as W


def test_known_answers():
    idx, payload = W.make_wire_golden()
    assert payload.tolist() == [[1, 8, 48, 0, 1, 5, 24, 112, 0, 2]]       # sum_q (q+1) << 10q, little-endian
    assert np.array_equal(W.unpack(payload, 8, 10, 1024), idx)
    assert W.frame_bytes(8, 10) == 10 and W.frame_bytes(16, 10) == 20 and W.frame_bytes(3, 10) == 4


def test_round_trip_property():
    rng = np.random.default_rng(1)
    for n_q, size, bits in ((8, 1024, 10), (16, 1024, 10), (5, 300, 9), (1, 2, 1)):
        idx = rng.integers(0, size, (n_q, 64)) + size * np.arange(n_q)[:, None]
        p = W.pack(idx, bits, size)
        assert p.shape == (64, W.frame_bytes(n_q, bits))
        assert np.array_equal(W.unpack(p, n_q, bits, size), idx)
    # padding bits of the last byte are zero
    p = W.pack(np.array([[1], [2 + 4], [3 + 8]]), 2, 4)
    assert p.tolist() == [[0b00111001]]
