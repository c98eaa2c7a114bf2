"""GPU: the DEFERRED guard of audiodec_amd.pipeline.StreamingPipeline -- the default mode of bench.py's timed region.

256 streams of vctk_v1 through the three-stream schedule with AudioDec's default guard: nothing synchronises per step, the flag
words of every program step are posted behind it and read one to `depth` batches late.  A stream is driven beyond the f16 range
in the ENCODER (audio x 1e6 for one frame) and, later, another one in the VOCODER (its zq x 1e5 for one frame): the batches in
flight are rewound, the program concerned continues on its exact-f32 twin, the batches are repeated into the tensors the caller
holds -- no exception, one RuntimeWarning per repair, every frame of every stream within tolerance of the CPU oracle (exact f32
throughout; reference semantics: layers/conv_layer.py:153-156, 194-197, layers/vq_module.py:90-104 through oracle/audiodec_oracle.py),
the streams that did not overflow undisturbed."""
import os
import warnings

import numpy as np
import pytest
import torch

from audiodec_amd import native, synth
from audiodec_amd.pipeline import StreamingPipeline
from test_gpu_parity import load_audiodec, DEV
from test_oracle_golden import build_oracle_shared_warmup

pytestmark = pytest.mark.gpu
HOP = 300


def _run(ad, pipe, B, steps, seed, bad_x, bad_q, magic, frames=1):
    """bad_x = (batch, stream, factor); bad_q = (stream, factor): the zq of `stream` is scaled wherever its first emitted index
    equals `magic` -- a pure function of the codes, applied on the device without synchronising, so the repeat of a batch sees it too."""
    audio = [np.stack([synth.synth_audio(seed + j, s, frames * HOP) for s in range(B)]) for j in range(steps)]
    if bad_x is not None:
        audio[bad_x[0]][bad_x[1]] *= bad_x[2]
    xs = [torch.from_numpy(a)[:, None, :].to(DEV) for a in audio]
    orig_lookup = ad.rx_encoder.lookup

    def lookup(idx):
        zq = orig_lookup(idx)
        if bad_q is not None:
            i3 = idx if idx.dim() == 3 else idx.unsqueeze(1)
            f = torch.where(i3[0, bad_q[0], 0] == magic, bad_q[1], 1.0).to(zq.dtype)
            zq[bad_q[0]] *= f
        return zq
    ad.rx_encoder.lookup = lookup
    ys, zs, idxs, warned = [], [], [], []
    try:
        with torch.no_grad():
            torch.cuda.synchronize()
            pipe.enter()
            for j in range(steps):
                with warnings.catch_warnings(record=True) as w:
                    warnings.simplefilter("always")
                    ys.append(pipe.step(xs[j]))
                warned.append(sum(issubclass(i.category, RuntimeWarning) for i in w))
                zs.append(pipe.last_z); idxs.append(pipe.last_idx)
            with warnings.catch_warnings(record=True) as w:
                warnings.simplefilter("always")
                pipe.exit()
            warned.append(sum(issubclass(i.category, RuntimeWarning) for i in w))
            torch.cuda.synchronize()
    finally:
        ad.rx_encoder.lookup = orig_lookup
    return xs, [t.cpu() for t in zs], [t.cpu() for t in idxs], [t.cpu() for t in ys], warned


@pytest.mark.parametrize("model,B,stages,frames,max_frames", [("vctk_v1", 256, "2", 1, 1), ("vctk_sym", 8, "1", 1, 1), ("vctk_sym", 5, "1", 3, 2)],
                         ids=["vctk_v1-256-2", "vctk_sym-8-1", "vctk_sym-5-three_frames_in_chunks_of_two"])
def test_deferred_guard_repairs_overflows_in_the_pipelined_schedule(gpu, ckpt_root, model, B, stages, frames, max_frames):
    """(third case: three frames per batch through programs that take two per step -- every batch is two program steps per program, of
    2 and 1 frames, each with its own post; a repair rewinds them one by one)"""
    steps, seed = 9, 4242
    old = os.environ.get("ADK_VOCODER_STAGES")
    os.environ["ADK_VOCODER_STAGES"] = stages
    try:
        ad = load_audiodec(ckpt_root, model, seed, B, max_frames, True)      # guard=None: the default, on
    finally:
        if old is None:
            del os.environ["ADK_VOCODER_STAGES"]
        else:
            os.environ["ADK_VOCODER_STAGES"] = old
    assert ad.tx_encoder.guard and ad.decoder.guard and ad.tx_encoder.split16 and ad.tx_encoder.rewind_depth >= 4
    pipe = StreamingPipeline(ad, DEV)
    assert pipe.deferred and pipe.depth == 4
    enc_prog = ad.tx_encoder._encoder()
    dec_progs = ad.decoder._decoder_stages() if hasattr(ad.decoder, "_decoder_stages") else [ad.decoder._decoder()]

    # the oracle first: it says which code marks the batch whose zq is scaled
    tx, rx, dec = build_oracle_shared_warmup(model, B, seed)
    bad_x, q_stream, q_batch = (2, 1, 1e6), 2, 5
    audio = [np.stack([synth.synth_audio(seed + j, s, frames * HOP) for s in range(B)]) for j in range(steps)]
    audio[bad_x[0]][bad_x[1]] *= bad_x[2]
    oz, oi, oy = [], [], []
    with torch.no_grad():
        for j in range(steps):
            z_ = tx.encode(torch.from_numpy(audio[j])[:, None, :])
            oz.append(z_); oi.append(tx.quantize(z_))
    magic = int(oi[q_batch].reshape(oi[q_batch].shape[0], B, -1)[0, q_stream, 0])
    with torch.no_grad():
        for j in range(steps):
            zq = rx.lookup(oi[j])
            i3 = oi[j].reshape(oi[j].shape[0], B, -1)
            if int(i3[0, q_stream, 0]) == magic:
                zq = zq.clone(); zq[q_stream] *= 1e5
            oy.append(dec.decode(zq))

    hits = [j for j in range(steps) if int(oi[j].reshape(oi[j].shape[0], B, -1)[0, q_stream, 0]) == magic]
    q_first = hits[0]                                          # (the marked code may occur in an earlier batch too: the rule is a function of the codes)

    xs, zs, idxs, ys, warned = _run(ad, pipe, B, steps, seed, bad_x, (q_stream, 1e5), magic, frames)
    # the encoder overflowed in batch 2, the vocoder in batch q_first: one or two repairs (two overflows inside one window of unverified
    # batches are one repair), each announced at a LATER step or at exit() -- never in the step that issued the bad batch
    assert 1 <= pipe.log.repairs <= 2 and sum(warned) >= pipe.log.repairs, (pipe.log.repairs, warned)
    assert warned[min(bad_x[0], q_first)] == 0 and sum(warned[:min(bad_x[0], q_first) + 1]) == 0
    assert enc_prog.demoted and not enc_prog.split16 and any(p.demoted for p in dec_progs)
    assert pipe.log.verified == steps and not pipe.log.pending
    assert native.device_flags() == 0
    clean = torch.ones(B, dtype=torch.bool)                    # streams whose codes have matched the oracle's so far decode the same signal
    for j in range(steps):
        dz = (zs[j] - oz[j]).abs().amax(dim=(1, 2)) / oz[j].abs().amax(dim=(1, 2)).clamp(min=1.0)
        tol = torch.full((B,), 1e-4); tol[bad_x[1]] = 1e-4 if j < bad_x[0] else 1e-3      # (1e6-sized samples stay in that stream's state for a receptive field)
        assert bool((dz < tol).all()), (j, float(dz.max()), int(dz.argmax()))
        same = (idxs[j].reshape(oi[j].shape) == oi[j]).reshape(oi[j].shape[0], B, -1).all(0).all(-1)
        clean &= same
        others = clean.clone(); others[bad_x[1]] = True        # (the overdriven stream's codes may sit on either side of a tie from batch 2 on)
        assert bool(others.all()) and (j >= bad_x[0] or bool(clean.all())), (j, (~others).nonzero().flatten().tolist())
        dy = (ys[j] - oy[j]).abs().amax(dim=(1, 2)) / oy[j].abs().amax(dim=(1, 2)).clamp(min=1.0)
        tol = torch.full((B,), 1e-4); tol[q_stream] = 1e-3 if j >= q_first else 1e-4
        assert bool(torch.isfinite(ys[j]).all()) and bool((dy < tol)[clean].all()), (j, float(dy[clean].max()), int(dy.argmax()))


def test_deferred_guard_is_bit_identical_to_the_unguarded_schedule_when_nothing_overflows(gpu, ckpt_root):
    """Same streams, same schedule, guard deferred vs guard off: the extra ring rows and the posts change no result bit."""
    B, steps, seed = 64, 6, 99
    outs = []
    for guard in (None, False):
        ad = load_audiodec(ckpt_root, "vctk_v1", seed, B, 1, True, guard=guard)
        pipe = StreamingPipeline(ad, DEV)
        assert pipe.deferred == (guard is None)
        xs, zs, idxs, ys, warned = _run(ad, pipe, B, steps, seed, None, None, -1)
        assert sum(warned) == 0
        if pipe.log is not None:
            assert pipe.log.verified == steps and pipe.log.repairs == 0
        outs.append((zs, idxs, ys))
    for a, b in zip(outs[0], outs[1]):
        for u, v in zip(a, b):
            assert torch.equal(u, v)
    assert native.device_flags() == 0


def test_replay_step_skips_the_ring_writes(gpu, ckpt_root):
    """ADK_STEP_REPLAY: after rewind() a step repeated WITHOUT its input buffer (the caller may have overwritten it) gives the same
    output, because the input rows are still in the program's first ring."""
    B, seed = 3, 7
    ad = load_audiodec(ckpt_root, "vctk_sym", seed, B, 1, True, guard=False)
    enc = ad.tx_encoder
    x = torch.from_numpy(np.stack([synth.synth_audio(seed, s, 3 * HOP) for s in range(B)]))[:, None, :].to(DEV)
    with torch.no_grad():
        z0 = enc.encode(x[:, :, :HOP]).clone()
        z1 = enc.encode(x[:, :, HOP:2 * HOP]).clone()
        prog = enc._encoder()
        prog.rewind(1); prog.rewind(1)
        enc._replay = True
        try:
            garbage = torch.full_like(x[:, :, :HOP], 123.0)
            r0 = enc.encode(garbage).clone()
            r1 = enc.encode(garbage).clone()
        finally:
            enc._replay = False
        z2 = enc.encode(x[:, :, 2 * HOP:]).clone()
    assert torch.equal(z0, r0) and torch.equal(z1, r1)
    ad2 = load_audiodec(ckpt_root, "vctk_sym", seed, B, 1, True, guard=False)
    with torch.no_grad():
        for k in range(3):
            w = ad2.tx_encoder.encode(x[:, :, k * HOP:(k + 1) * HOP])
    assert torch.equal(w, z2)
    assert native.device_flags() == 0
