"""GPU: the guard of direct calls in its default ("lazy") mode -- audiodec_amd/lazy_guard.py -- through the drop-in surface
(AudioDec.tx_encoder.encode / quantize, rx_encoder.lookup, decoder.decode; /root/reference/utils/audiodec.py:100-106)."""
import warnings

import numpy as np
import pytest
import torch

from audiodec_amd import lazy_guard, synth
from test_gpu_parity import DEV, build_oracle, load_audiodec

pytestmark = pytest.mark.gpu
HOP = 300


def _step(ad, x):
    z = ad.tx_encoder.encode(x)
    idx = ad.tx_encoder.quantize(z)
    zq = ad.rx_encoder.lookup(idx)
    return z, idx, ad.decoder.decode(zq)


def test_lazy_and_synchronous_guard_agree_bit_for_bit_and_lazy_calls_do_not_wait(gpu, ckpt_root):
    """Nothing overflows: the same calls with the check deferred (default) and with one stream synchronisation per program step give the
    same bits; the lazy calls never wait while their results are only handed on to the next call, and looking at a result settles the log."""
    B, steps = 5, 6
    audio = np.stack([synth.synth_audio(7, s, steps * HOP) for s in range(B)])
    ad_l = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 1, True)
    ad_s = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 1, True)
    for g in (ad_s.tx_encoder, ad_s.rx_encoder, ad_s.decoder):
        g.set_guard(True, "sync")
    log = ad_l.tx_encoder._log
    assert log is not None and log is ad_l.decoder._log and log is ad_l.rx_encoder._log and ad_l.tx_encoder.guard_mode == "lazy"
    outs = []
    ad_l.settle()
    v0 = log.verified                                           # (the warm-up calls)
    with torch.no_grad():
        for f in range(4):                                      # within the rings' rewind depth: nobody waits, nothing is looked at
            x = torch.from_numpy(audio[:, f * HOP:(f + 1) * HOP])[:, None, :].to(DEV)
            outs.append(_step(ad_l, x))
        assert all(type(t) is lazy_guard.GuardedTensor for o in outs for t in o)
        assert log.waits == 0 and len(log.pending) > 0
        for f in range(4, steps):                               # beyond it: a call waits for the oldest one, nothing else
            x = torch.from_numpy(audio[:, f * HOP:(f + 1) * HOP])[:, None, :].to(DEV)
            outs.append(_step(ad_l, x))
        y_last = outs[-1][2].cpu()                              # the first look settles everything
        assert not log.pending and log.repairs == 0 and log.verified - v0 == 4 * steps
        for f in range(steps):
            x = torch.from_numpy(audio[:, f * HOP:(f + 1) * HOP])[:, None, :].to(DEV)
            zs, idxs, ys = _step(ad_s, x)
            assert type(ys) is torch.Tensor
            assert torch.equal(outs[f][0].cpu(), zs.cpu()) and torch.equal(outs[f][1].cpu(), idxs.cpu()) and torch.equal(outs[f][2].cpu(), ys.cpu()), f
    assert torch.isfinite(y_last).all()
    from audiodec_amd import native
    assert native.device_flags() == 0


def test_an_overflow_found_late_repairs_every_call_that_consumed_it(gpu, ckpt_root):
    """Frame 1 of stream 1 carries 1e6-sized samples: the encoder's split-f16 step overflows.  Nothing is looked at while two more frames go
    through encode -> quantize -> lookup -> decode on top of it (garbage in, garbage out, all unverified) -- the overflow is found by a later
    call's poll or, at the latest, by the first look at a result.  ONE repair: the encoder continues on its exact-f32 twin, and z, the indices
    and the waveform of EVERY frame, in the tensors handed out before the repair, are those of the CPU oracle."""
    B, steps, seed, model = 3, 4, 1337, "vctk_sym"
    ad = load_audiodec(ckpt_root, model, seed, B, 1, True)
    tx, rx, dec = build_oracle(model, B, seed)
    audio = np.stack([synth.synth_audio(55, s, steps * HOP) for s in range(B)])
    xs = []
    for f in range(steps):
        x = torch.from_numpy(audio[:, f * HOP:(f + 1) * HOP].copy())[:, None, :]
        if f == 1:
            x[1] *= 1e6
        xs.append(x)
    log = ad.tx_encoder._log
    outs = []
    with torch.no_grad():
        with warnings.catch_warnings(record=True) as w:
            warnings.simplefilter("always")
            for f in range(steps):
                xd = xs[f].to(DEV)
                outs.append(_step(ad, xd))
                xd.zero_()                                      # the caller reuses its input tensor: the repeat must not need it (ADK_STEP_REPLAY)
            y3 = outs[3][2].cpu()
        assert any(issubclass(i.category, RuntimeWarning) for i in w) and log.repairs == 1 and not log.pending
        assert ad.tx_encoder._encoder().demoted and not ad.decoder._decoder().demoted
        for f in range(steps):
            oz = tx.encode(xs[f])
            oi = tx.quantize(oz)
            oy = dec.decode(rx.lookup(oi))
            z, idx, y = (t.cpu() for t in outs[f])
            dz = (z - oz).abs().amax(dim=(1, 2)) / oz.abs().amax(dim=(1, 2)).clamp(min=1.0)
            tol = torch.full((B,), 1e-4); tol[1] = 1e-4 if f <= 1 else 1e-3          # (1e6-sized state for a receptive field: as the synchronous test)
            assert bool((dz < tol).all()), (f, dz)
            for s in (0, 2):                                     # the undisturbed streams: exact codes, waveform within the north-star's tolerance
                assert torch.equal(idx[:, s], oi[:, s]), (f, s)
                assert float((y[s] - oy[s]).abs().max()) < 1e-4, (f, s)
            assert bool(torch.isfinite(y).all())
    assert torch.equal(y3, outs[3][2].cpu())
    from audiodec_amd import native
    assert native.device_flags() == 0
