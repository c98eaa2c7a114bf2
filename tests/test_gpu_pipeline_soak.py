"""Soak of the SHIPPED schedule: what bench.py times, run long.

256 streams of `vctk_v1`, one frame per stream per step, >= 500 steps through bench.TxRxPipeline -- three HIP streams with three
batches in flight, the residual chains as one launch each (conv_rb16: inline-asm LDS-DMA, hand-managed M0, bare s_barrier), the
conv_out + transposed-conv launch (conv_ou16), nothing synchronising between steps (guard=False).  Three claims:

  1. the run is REPRODUCIBLE: a second, fresh model fed the same audio emits bit-identical latents, indices and waveforms for
     all 256 streams over all steps (a race in the chain kernels, an LDS-DMA that lands late one time in 10^4, a lost hand-over
     between HIP streams would show here), and the device flag words stay 0;
  2. it is CORRECT over the whole run, not only over the first steps: 8 sampled streams are compared with per-stream oracles
     (B reference instances, oracle/audiodec_oracle.py) at every step -- waveform <= 1e-4 max-abs, indices bit-exact (a flip is
     accepted only where the reference's own top-2 margin is below 1e-4, i.e. within reach of the latent's f32 round-off, and
     from there on that stream's waveform is not compared: it decodes other codes);
  3. the RVQ soak of test_gpu_soak.py also holds behind the chain-kernel encoder: 256 streams x 50 single-frame steps x 8 stages
     = 102,400 decisions (with 8-frame chunks at 64 streams, as there, the product runs the encoder op by op).

Reference semantics: CausalConv1d / CausalConvTranspose1d.inference (layers/conv_layer.py:153-156, 194-197),
HiFiGANResidualBlock.inference (models/vocoder/modules/residual_block.py:99-105), CausalResidualUnit.inference
(models/autoencoder/modules/residual_unit.py:78-81), ResidualVQ.forward_index (layers/vq_module.py:136-149).
ADK_SOAK_STEPS shortens the run for local experiments (the suite's value is 500).
"""
import os

import numpy as np
import pytest
import torch

from audiodec_amd import native, synth
from test_gpu_parity import load_audiodec, DEV, WAVE_TOL
from test_oracle_golden import build_oracle_shared_warmup
from test_gpu_soak import _first_flips, _report, BOUND_END_TO_END

pytestmark = pytest.mark.gpu

HOP = 300
B = 256
STEPS = int(os.environ.get("ADK_SOAK_STEPS", "500"))
SAMPLED = [0, 1, 37, 100, 128, 200, 254, 255]
SEED_AUDIO = 31337


def _chain_kernels(ad):
    progs = [ad.tx_encoder._encoder()] + list(ad.decoder._decoder_stages())
    return {pr.describe_op(i, 1) for pr in progs for i in range(pr.n_ops) if pr._ops[i].kind == 0}


def _pipeline_run(ckpt_root, xs, steps):
    """A fresh model through bench.TxRxPipeline: (z, idx, y) of all streams and steps, on the device."""
    import bench
    old = os.environ.get("ADK_VOCODER_STAGES")
    os.environ["ADK_VOCODER_STAGES"] = "2"                      # bench.py's default lowering
    try:
        ad = load_audiodec(ckpt_root, bench.MODEL, bench.SEED, B, 1, True)         # the default guard, deferred by the pipeline: what bench.py times
    finally:
        if old is None:
            del os.environ["ADK_VOCODER_STAGES"]
        else:
            os.environ["ADK_VOCODER_STAGES"] = old
    names = _chain_kernels(ad)
    assert {"conv_rb16<32>", "conv_rb16<64>", "conv_rb16<128>", "conv_ou16<192>", "conv_sk16<64x64>"} <= names, names
    pipe = bench.TxRxPipeline(ad, DEV)
    assert len(pipe._all()) == 3 and pipe.deferred
    zs, idxs, ys = [], [], []
    with torch.no_grad():
        torch.cuda.synchronize()
        pipe.enter()
        for j in range(steps):
            ys.append(pipe.step(xs[j]))
            zs.append(pipe.last_z); idxs.append(pipe.last_idx)
        pipe.exit()
        torch.cuda.synchronize()
    assert native.device_flags() == 0
    assert pipe.log.verified == steps and pipe.log.repairs == 0
    return torch.cat(zs, -1), torch.cat(idxs, -1), torch.cat(ys, -1)


@pytest.fixture(autouse=True)
def _bounded_oracle_threads():
    """The oracle's ATen CPU convs are small: with every core of a 256-cpu GPU box in the intra-op pool a frame takes many times longer
    than with a handful (profiles/r2_cpu_legs.md)."""
    old = torch.get_num_threads()
    torch.set_num_threads(min(16, old))
    yield
    torch.set_num_threads(old)


def test_chain_pipeline_soak_is_reproducible_and_matches_per_stream_oracles(gpu, ckpt_root):
    audio = np.stack([synth.synth_audio(SEED_AUDIO, s, STEPS * HOP) for s in range(B)])
    xs = [torch.from_numpy(np.ascontiguousarray(audio[:, j * HOP:(j + 1) * HOP]))[:, None, :].to(DEV) for j in range(STEPS)]   # as bench.py: contiguous batches
    z1, i1, y1 = _pipeline_run(ckpt_root, xs, STEPS)
    z2, i2, y2 = _pipeline_run(ckpt_root, xs, STEPS)
    assert z1.shape == (B, 64, STEPS) and i1.shape == (8, B, STEPS) and y1.shape == (B, 1, STEPS * HOP)
    # 1. bit-identical reruns, every stream, every step
    assert torch.equal(i1, i2), f"indices differ between two runs at {int((i1 != i2).sum())} places"
    assert torch.equal(z1, z2), f"latents differ between two runs: max {float((z1 - z2).abs().max()):.3e}"
    assert torch.equal(y1, y2), f"waveforms differ between two runs: max {float((y1 - y2).abs().max()):.3e}"
    assert bool(torch.isfinite(y1).all())
    # 2. sampled streams against per-stream oracles over ALL steps
    import bench
    n = len(SAMPLED)
    tx, rx, dec = build_oracle_shared_warmup(bench.MODEL, n, bench.SEED)
    xa = torch.from_numpy(audio[SAMPLED])[:, None, :]
    z = z1[SAMPLED].cpu().numpy(); idx = i1[:, SAMPLED].cpu().numpy(); y = y1[SAMPLED].cpu().numpy()
    oz, oi, om, oy = [], [], [], []
    with torch.no_grad():
        for j in range(STEPS):                                  # the same call sequence: one frame per call
            z_ = tx.encode(xa[:, :, j * HOP:(j + 1) * HOP])
            i_, m_ = tx.quantize(z_, return_margin=True)
            oy.append(dec.decode(rx.lookup(i_))); oz.append(z_); oi.append(i_); om.append(m_)
    oz = torch.cat(oz, -1).numpy(); oi = torch.cat(oi, -1).numpy(); om = torch.cat(om, -1).numpy(); oy = torch.cat(oy, -1).numpy()
    dz = float(np.abs(z - oz).max())
    flips = _first_flips(idx, oi, om)
    # a stream whose codes differed at frame t decodes a different signal from then on: compare its waveform up to there
    upto = np.full(n, STEPS)
    for b, t, _, _ in flips:
        upto[b] = min(upto[b], t)
    dy = max(float(np.abs(y[b, :, :upto[b] * HOP] - oy[b, :, :upto[b] * HOP]).max()) if upto[b] else 0.0 for b in range(n))
    _report("chain_pipeline", flips, om, {"max_abs_dz": dz, "max_abs_dy": dy, "streams": B, "steps": STEPS, "sampled_streams": SAMPLED,
                                          "frames_compared_per_sampled_stream": [int(u) for u in upto], "bound": BOUND_END_TO_END,
                                          "reruns_bit_identical": True})
    assert dz < WAVE_TOL, dz
    worst = max((m for *_, m in flips), default=0.0)
    assert worst < BOUND_END_TO_END, f"{len(flips)} flips, largest reference margin {worst:.3e}"
    assert int(upto.min()) >= STEPS // 2, f"a sampled stream left the comparison early: {upto}"
    assert dy < WAVE_TOL, f"max|dy| = {dy:.3e}"


def test_rvq_soak_behind_the_chain_kernel_encoder(gpu, ckpt_root):
    """256 streams x 50 single-frame steps x 8 stages = 102,400 decisions; the encoder runs its residual chains as conv_rb16."""
    frames = 50
    audio = np.stack([synth.synth_audio(2025, s, frames * HOP) for s in range(B)])
    tx, _, _ = build_oracle_shared_warmup("vctk_v1", B, 1337)
    ad = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 1, True)
    enc = ad.tx_encoder._encoder()
    names = {enc.describe_op(i, 1) for i in range(enc.n_ops) if enc._ops[i].kind == 0}
    assert {"conv_rb16<32>", "conv_rb16<64>", "conv_rb16<128>"} <= names, names
    zs, idxs, ozs, ois, oms = [], [], [], [], []
    with torch.no_grad():
        for f in range(frames):
            x = torch.from_numpy(audio[:, f * HOP:(f + 1) * HOP])[:, None, :]
            z = ad.tx_encoder.encode(x.to(DEV))
            idxs.append(ad.tx_encoder.quantize(z).cpu()); zs.append(z.cpu())
            oz = tx.encode(x)
            oi, om = tx.quantize(oz, return_margin=True)
            ozs.append(oz); ois.append(oi); oms.append(om)
    z = torch.cat(zs, -1); idx = torch.cat(idxs, -1).numpy()
    oz = torch.cat(ozs, -1); oi = torch.cat(ois, -1).numpy(); om = torch.cat(oms, -1).numpy()
    assert idx.shape == oi.shape == (8, B, frames) and idx.size >= 100_000
    dz = float((z - oz).abs().max())
    flips = _first_flips(idx, oi, om)
    rep = _report("end_to_end_chain_encoder_split16", flips, om, {"max_abs_dz": dz, "bound": BOUND_END_TO_END, "streams": B, "frames": frames})
    assert dz < 1e-4, dz
    worst = max((m for *_, m in flips), default=0.0)
    assert worst < BOUND_END_TO_END, f"{len(flips)} flips, largest reference margin {worst:.3e}: {rep['flip_list'][:5]}"
