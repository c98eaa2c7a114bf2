"""Seeded inputs for the layer-level known-answer fixtures (tests/golden/ops.npz).

Inputs and weights are regenerated from a CPU torch.Generator (deterministic across machines);
only the reference's OUTPUTS are stored in the fixture (see make_golden.py:op_cases).
"""
import torch

SEED = 1337

# (cin, cout, k, stride, dilation, groups, bias, L1 (out frames chunk 1), L2)
CONVS = [
    (32, 32, 7, 1, 9, 1, False, 30, 12),      # AE residual-unit conv, dilation 9
    (32, 64, 6, 3, 1, 1, True, 30, 12),       # encoder down-sampling conv K=2s
    (96, 96, 11, 1, 5, 3, True, 20, 7),       # vocoder grouped K11 d5
    (1, 32, 7, 1, 1, 1, False, 40, 10),       # encoder input conv
    (32, 1, 7, 1, 1, 1, True, 40, 10),        # output conv
    (64, 32, 3, 1, 1, 1, False, 1, 2),        # projector-like K3, single-frame chunks
    (64, 64, 7, 1, 3, 1, False, 3, 50),       # chunk shorter than the history, then longer
]
# (cin, cout, stride, L1, L2)
CONVTS = [(64, 32, 3, 10, 4), (128, 64, 4, 5, 1), (32, 32, 5, 1, 3), (32, 32, 2, 6, 2)]

RVQ_STAGES, RVQ_ROWS = 4, 50


def conv_inputs(n):
    ci, co, k, s, d, gr, b, L1, L2 = CONVS[n]
    g = torch.Generator().manual_seed(SEED + 17 * n)
    x1 = torch.randn(1, ci, L1 * s, generator=g)
    x2 = torch.randn(1, ci, L2 * s, generator=g)
    w = torch.randn(co, ci // gr, k, generator=g) / (ci // gr * k) ** 0.5
    bias = torch.randn(co, generator=g) * 0.1 if b else None
    return x1, x2, w, bias


def convt_inputs(n):
    ci, co, s, L1, L2 = CONVTS[n]
    g = torch.Generator().manual_seed(SEED + 1000 + 17 * n)
    x1 = torch.randn(1, ci, L1, generator=g)
    x2 = torch.randn(1, ci, L2, generator=g)
    w = torch.randn(ci, co, 2 * s, generator=g) / (2 * ci) ** 0.5
    bias = torch.randn(co, generator=g) * 0.1
    return x1, x2, w, bias


def rvq_inputs():
    g = torch.Generator().manual_seed(SEED + 2000)
    embeds = [torch.randn(64, 1024, generator=g) * (0.8 ** i) for i in range(RVQ_STAGES)]
    embeds[0][:, 777] = embeds[0][:, 123]          # duplicate code -> exact tie, lowest index must win
    x = torch.randn(1, RVQ_ROWS, 64, generator=g)
    x[0, 7] = embeds[0][:, 123]                    # a row sitting exactly on the tie
    return embeds, x
