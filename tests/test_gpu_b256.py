"""GPU parity in the configuration bench.py TIMES: vctk_v1, 256 streams, one frame per stream per step, the
kernel choices AUTO makes at that size (rows-in-LDS split kernels, 128x64 stream-K tiles), bench.py's own
schedule object (three HIP streams, two vocoder programs, 256 persistent workgroups per concurrent program).

Reference semantics checked: CausalConv1d / CausalConvTranspose1d.inference per stream
(layers/conv_layer.py:153-156, 194-197) and ResidualVQ.forward_index (layers/vq_module.py:90-104, 136-149),
through the B-stream oracle (B reference instances; oracle/audiodec_oracle.py).  Tolerances: waveform and latent
<= 1e-4 max-abs, indices bit-exact (a flip is reported with the reference's own top-2 margin).
"""
import os

import numpy as np
import pytest
import torch
import torch.nn.functional as F

from audiodec_amd import synth
from test_gpu_parity import load_audiodec, DEV, WAVE_TOL
from test_oracle_golden import build_oracle_shared_warmup, explain_flips

pytestmark = pytest.mark.gpu

HOP = 300


def _program_kernels(ad):
    """(op name, kernel, taps, has residual) of every conv op of the transmitter and receiver programs at 1 frame/step."""
    progs = [ad.tx_encoder._encoder()] + list(ad.decoder._decoder_stages())
    out = []
    for pr in progs:
        for i in range(pr.n_ops):
            op = pr._ops[i]
            if op.kind == 0:
                out.append((pr.op_names[i], pr.describe_op(i, 1), op.conv.taps, op.res_ring >= 0))
    return out


@pytest.mark.parametrize("split16", [True, False], ids=["split16", "f32"])
def test_benched_configuration_matches_oracle(gpu, ckpt_root, split16):
    """256 streams x 4 steps through bench.py's TxRxPipeline; every step's z, indices and waveform against the oracle."""
    import bench
    B, steps, seed = 256, 4, bench.SEED
    old = os.environ.get("ADK_VOCODER_STAGES")
    os.environ["ADK_VOCODER_STAGES"] = "2"                     # bench.py --stages 2 (its default)
    try:
        ad = load_audiodec(ckpt_root, bench.MODEL, seed, B, 1, split16)      # as bench.py: AudioDec's default guard, deferred by the pipeline object
    finally:
        if old is None:
            del os.environ["ADK_VOCODER_STAGES"]
        else:
            os.environ["ADK_VOCODER_STAGES"] = old
    assert ad.decoder.stages == 2
    pipe = bench.TxRxPipeline(ad, DEV)                         # streams / workgroup share exactly as bench.py sets them up
    assert pipe.deferred and pipe.depth == 4                   # nothing synchronises between steps; every program step is checked one to four batches late
    assert ad.decoder.workgroups == 0 and ad.tx_encoder.workgroups == 0      # (library default since round 2: no cap)
    kern = _program_kernels(ad)
    names = {k for _, k, _, _ in kern}
    if split16:
        # the kernels bench.py's headline number is made of -- all of them are in the programs checked here
        assert {"conv_rb16<32>", "conv_rb16<64>", "conv_rb16<128>", "conv_sk16<64x64>", "conv_ou16<192>"} <= names, names
        by_name = dict((n, k) for n, k, _, _ in kern)
        # the north-star's named kernel (LeakyReLU -> ConvTranspose1d 64 -> 32, s3), with the 1x1 conv_out that feeds it pulled in
        assert by_name["blocks.2.conv_out"] == "conv_ou16<192>" and by_name["upsamples.3"] == "(fused into the previous op)"
        # every residual chain with 32 / 64 / 128 channels per group runs as ONE launch: the three residual units of encoder blocks
        # 0-2 (6 convs each) and the residual blocks of vocoder stages 1-3 (6 grouped K11 convs each)
        for head, C_ in (("encoder.conv_blocks.0.res_units.0.conv1", 32), ("encoder.conv_blocks.1.res_units.0.conv1", 64),
                         ("encoder.conv_blocks.2.res_units.0.conv1", 128), ("blocks.1.convs1.0", 128), ("blocks.2.convs1.0", 64),
                         ("blocks.3.convs1.0", 32)):
            assert by_name[head] == f"conv_rb16<{C_}>", (head, by_name[head])
        for tail in ("encoder.conv_blocks.0.res_units.0.conv2", "encoder.conv_blocks.2.res_units.2.conv2", "blocks.1.convs2.2", "blocks.3.convs1.1"):
            assert by_name[tail] == "(fused into the previous op)", (tail, by_name[tail])
        # the wide grouped convs of stage 0 and encoder block 3 read shadow rings on the stream-K kernel
        assert by_name["blocks.0.convs1.0"] == "conv_sk16<64x64>" and by_name["encoder.conv_blocks.3.res_units.0.conv1"] == "conv_sk16<64x64>"
        assert sum(1 for _, k, _, _ in kern if k != "(fused into the previous op)") <= 45 - 5 - 1  # launches per step incl. ring writes, RVQ, lookup
    else:
        assert {"conv_rl<32>", "conv_rl<64>", "conv_sk<64x64>"} <= names, names
    # bench.py's inputs: stream s of batch j = synth_audio(SEED + j, s, HOP)
    xs = [torch.from_numpy(np.stack([synth.synth_audio(seed + j, s, HOP) for s in range(B)]))[:, None, :].to(DEV) for j in range(steps)]
    zs, idxs, ys = [], [], []
    with torch.no_grad():
        torch.cuda.synchronize()
        pipe.enter()
        for j in range(steps):
            ys.append(pipe.step(xs[j]))
            zs.append(pipe.last_z); idxs.append(pipe.last_idx)
        pipe.exit()
        torch.cuda.synchronize()
    assert pipe.log.verified == steps and pipe.log.repairs == 0
    z = torch.cat([t.cpu() for t in zs], -1).numpy()
    idx = torch.cat([t.cpu() for t in idxs], -1).numpy()
    y = torch.cat([t.cpu() for t in ys], -1).numpy()
    from audiodec_amd import native
    assert native.device_flags() == 0

    tx, rx, dec = build_oracle_shared_warmup(bench.MODEL, B, seed)
    oz, oi, om, oy = [], [], [], []
    with torch.no_grad():
        for j in range(steps):
            z_ = tx.encode(xs[j].cpu())
            i_, m_ = tx.quantize(z_, return_margin=True)
            oy.append(dec.decode(rx.lookup(i_))); oz.append(z_); oi.append(i_); om.append(m_)
    oz = torch.cat(oz, -1).numpy(); oi = torch.cat(oi, -1).numpy(); om = torch.cat(om, -1).numpy(); oy = torch.cat(oy, -1).numpy()
    assert z.shape == oz.shape and idx.shape == oi.shape and y.shape == oy.shape == (B, 1, steps * HOP)
    assert np.abs(z - oz).max() < WAVE_TOL, f"max|dz| = {np.abs(z - oz).max():.3e}"
    explain_flips(idx, oi, om, f"{bench.MODEL} B={B} bench schedule")
    assert np.abs(y - oy).max() < WAVE_TOL, f"max|dy| = {np.abs(y - oy).max():.3e}"


# ------------------------------------------------------------------------------------------------
# op level: the two kernel variants no layer fixture reaches
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("C_,T,B,bias", [(32, 300, 5, False), (64, 100, 4, False), (32, 77, 3, True), (64, 33, 7, True)])
def test_rows_kernel_1x1_with_residual_matches_oracle(gpu, C_, T, B, bias):
    """CausalResidualUnit.inference's second half, y = x + conv2(ELU(h)) (residual_unit.py:78-81: 1x1 conv, no bias in the
    reference; a bias variant is checked too), in the split rows-in-LDS kernel with the fused residual epilogue."""
    from audiodec_amd import layers, native
    g = torch.Generator().manual_seed(1000 + C_ + T)
    w = torch.randn(C_, C_, 1, generator=g) / C_ ** 0.5
    b = torch.randn(C_, generator=g) * 0.1 if bias else None
    for impl, want in ((native.IMPL_SPLIT16_ROWS, f"conv_rl16<{C_}>"), (native.IMPL_SPLIT16_SK, None), (native.IMPL_MFMA, None)):
        m = layers.CausalConv1d(C_, C_, 1, 1, 1, 1, bias, device=gpu, batch=B, max_len=T).load(w, b)
        m.set_activation("ELU")
        m.impl = impl
        for step in range(3):
            h = torch.randn(B, C_, T, generator=g)
            x = torch.randn(B, C_, T, generator=g)
            ref = x + F.conv1d(F.elu(h), w, b)
            y = m.inference(h, residual=x).cpu()
            if want:
                assert m.last_kernel == want, m.last_kernel
            assert float((y - ref).abs().max()) < 2e-5, (impl, step, float((y - ref).abs().max()))
    assert native.device_flags() == 0


@pytest.mark.parametrize("d,with_res", [(1, False), (3, True), (5, True)])
@pytest.mark.parametrize("C_,T,B,impl_name,want", [(384, 25, 224, "SPLIT16", "conv_sk16<128x64>"), (768, 5, 67, "SPLIT16", "conv_sk16<64x64>")])
def test_deep_grouped_convs_match_oracle(gpu, d, with_res, C_, T, B, impl_name, want):
    """The grouped K11 convs of vocoder stages 0-1 (768 = 3 x 256 channels at 5 steps per frame, 384 = 3 x 128 at 25) at
    stream counts where the dispatch picks the 128-row stream-K tiles (conv_sk16<128x64>); LeakyReLU in, bias, residual epilogue (residual_block.py:99-105),
    column counts that are not a multiple of the tile, ring wrap-around over three steps."""
    from audiodec_amd import layers, native
    K, gr = 11, 3
    g = torch.Generator().manual_seed(77 + d)
    w = torch.randn(C_, C_ // gr, K, generator=g) / (C_ // gr * K) ** 0.5
    b = torch.randn(C_, generator=g) * 0.1
    m = layers.CausalConv1d(C_, C_, K, 1, d, gr, True, device=gpu, batch=B, max_len=T).load(w, b)
    m.set_activation("LeakyReLU", 0.1)
    m.impl = getattr(native, "IMPL_" + impl_name)                # SPLIT16 = AUTO among the split kernels, as the programs use it
    pad = torch.zeros(B, C_, (K - 1) * d)
    act = torch.nn.LeakyReLU(0.1)
    for step in range(3):
        x = torch.randn(B, C_, T, generator=g)
        r = torch.randn(B, C_, T, generator=g) if with_res else None
        xin = torch.cat((pad, x), -1)
        pad = xin[:, :, xin.shape[-1] - pad.shape[-1]:]
        ref = F.conv1d(act(xin), w, b, dilation=d, groups=gr)
        if with_res:
            ref = ref + r
        y = m.inference(x, residual=r).cpu()
        assert m.last_kernel == want, m.last_kernel
        assert float((y - ref).abs().max()) < 2e-5, (step, float((y - ref).abs().max()))
    assert native.device_flags() == 0


@pytest.mark.parametrize("cout,s,act,B,chunks", [(32, 3, "LeakyReLU", 9, [100, 100, 37, 100]), (32, 3, None, 3, [300, 5, 129]),
                                                 (16, 2, "ELU", 4, [64, 1, 200]), (24, 4, "LeakyReLU", 2, [50, 50]),
                                                 # >= 1024 chunks of 128 steps: two chunks per workgroup, the second prefetched (TPW = 2)
                                                 (32, 3, "LeakyReLU", 256, [500, 500, 389]),       # 5 frames per call at 256 streams; 4 chunks, the last ragged
                                                 (32, 3, "ELU", 8, [16500, 300, 16434]),           # 129 chunks per stream: the last pair has ONE chunk
                                                 (16, 2, None, 41, [3200, 3137])])                 # two m-tiles, 25 chunks x 41 streams = 1025
def test_upsampling_streamer_matches_oracle(gpu, cout, s, act, B, chunks):
    """conv_up16 (the north-star's named kernel: fused activation -> ConvTranspose1d(64 -> cout, K = 2s, stride s) + bias,
    models/vocoder/HiFiGAN.py:285-289, layers/conv_layer.py:194-197): chunk lengths that are not multiples of the 32-step
    tile, longer than one 128-step workgroup, one step, ring wrap-around; with and without the input activation (the
    symmetric decoder has none); also against the rows-in-LDS and stream-K kernels it replaces for this layer.  Launches of at least
    1024 chunks run two chunks per workgroup (second chunk's loads under the first chunk's MFMAs): even, odd and ragged chunk counts."""
    from audiodec_amd import layers, native
    from oracle import audiodec_oracle as O
    g = torch.Generator().manual_seed(31 * cout + s)
    w = torch.randn(64, cout, 2 * s, generator=g) / (2 * 64) ** 0.5
    bias = torch.randn(cout, generator=g) * 0.1
    fn = {None: lambda v: v, "ELU": torch.nn.ELU(), "LeakyReLU": torch.nn.LeakyReLU(0.1)}[act]
    mods = []
    for impl in (native.IMPL_SPLIT16_UP, native.IMPL_SPLIT16, native.IMPL_SPLIT16_SK, native.IMPL_MFMA):
        m = layers.CausalConvTranspose1d(64, cout, 2 * s, s, device=gpu, batch=B, max_len=max(chunks)).load(w, bias)
        m.set_activation(act, 0.1)
        m.impl = impl
        mods.append(m)
    pad = torch.zeros(B, 64, 1)
    for step, L in enumerate(chunks):
        x = torch.randn(B, 64, L, generator=g)
        ref, pad = O.causal_convtr1d_inference(fn(x), pad, w, bias, s)     # the reference's state holds the ACTIVATED input
        for m in mods:
            y = m.inference(x).cpu()
            if m.impl in (native.IMPL_SPLIT16_UP, native.IMPL_SPLIT16):
                assert m.last_kernel == "conv_up16<64>", m.last_kernel
            assert y.shape == ref.shape and float((y - ref).abs().max()) < 2e-5, (m.impl, step, float((y - ref).abs().max()))
    assert native.device_flags() == 0


def test_program_runs_on_a_device_that_is_not_current(gpu, ckpt_root):
    """A program is bound to the device of its buffers (include/audiodec_hip.h, 'devices'): with another device current
    (the reference's --tx_cuda / --rx_cuda split, demoStream.py:33-40) the launches, the stream-K workspace and the
    flag word still belong to the program's device.  With one GPU in the box the non-current case cannot be produced;
    the test then only checks the bound path end to end."""
    ad = load_audiodec(ckpt_root, "vctk_sym", 1337, 2, 2)
    x = torch.from_numpy(np.stack([synth.synth_audio(3, s, 600) for s in range(2)]))[:, None, :]
    with torch.no_grad():
        y0 = ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x.to(DEV))))).cpu()
    if torch.cuda.device_count() < 2:
        assert torch.isfinite(y0).all()
        return
    ad1 = None
    try:
        import test_gpu_parity as P
        old = P.DEV
        P.DEV = "cuda:1"
        ad1 = load_audiodec(ckpt_root, "vctk_sym", 1337, 2, 2)
    finally:
        P.DEV = old
    with torch.no_grad(), torch.cuda.device(0):                  # device 0 current, program on device 1
        y1 = ad1.decoder.decode(ad1.rx_encoder.lookup(ad1.tx_encoder.quantize(ad1.tx_encoder.encode(x.to("cuda:1"))))).cpu()
    assert torch.equal(y0, y1)
    from audiodec_amd import native
    assert native.device_flags() == 0


@pytest.mark.parametrize("ranks", [2, 8])
def test_bench_n_ranks_rehearsal_on_one_gpu(gpu, ranks):
    """`python bench.py --gpus N` as the driver types it: the script launches its own N ranks (torch.distributed.run, rendezvous
    on 127.0.0.1), the weights come from rank 0, the timing is barrier-bracketed and the max over ranks, rank 0 prints the one
    JSON line.  The ranks share this box's one GPU over gloo (ADK_BENCH_BACKEND=gloo ADK_BENCH_ONE_GPU=1); production is one
    rank per GPU over RCCL, and the 1 -> 8 GPU curve itself can only be measured by the driver on an 8-GPU node.  What CAN be
    rehearsed at N = 8 is the host side: eight Python ranks on one host, each pinned to its own eighth of the cpus
    (shard.pin_rank), each handing ~40 launches per step to the runtime -- the time a rank needs to issue one batch is recorded
    per rank and must stay below the 1-GPU step time (0.9 ms), i.e. the hosts would keep eight GPUs fed."""
    import json
    import subprocess
    import sys
    root = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))
    env = dict(os.environ, ADK_BENCH_BACKEND="gloo", ADK_BENCH_ONE_GPU="1")
    for k in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT", "LOCAL_WORLD_SIZE"):
        env.pop(k, None)
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", str(ranks), "--steps", "6", "--warmup", "2", "--preroll", "8", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=900, env=env, cwd=root)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                    # rank 0 only
    out = json.loads(lines[0])
    assert out["n_gpus"] == ranks and out["config"]["streams_total"] == 256 * ranks and out["config"]["streams_per_gpu"] == 256
    assert out["scaling"] == "weak" and out["steps"] == 6 and out["device_error_flags"] == 0
    assert out["value"] > 0 and abs(out["value"] - 256 * ranks * 6 / (out["ms_per_step"] * 6e-3)) < 0.01 * out["value"]
    assert out["roofline"]["kernel"].startswith("conv_") and 0 < out["roofline"]["frac"] < 1      # rank 0 prices its kernels at every N
    assert "cpu_baseline" not in out and "self_check" not in out                                   # single-GPU legs stay at N = 1
    assert len(out["frames_per_s_of_each_rank"]) == ranks
    host = out["host_ms_per_step_of_each_rank"]
    assert len(host["issue"]) == ranks and "pinned" in host["cpus_of_rank_0"], host
    assert max(host["issue"]) < 0.9, host                      # every rank issues a batch faster than one GPU finishes one
    if ranks == 8:
        rep = os.path.join(root, "gpurun_out", "r5_eight_ranks_one_gpu.json")
        os.makedirs(os.path.dirname(rep), exist_ok=True)
        json.dump({k: out[k] for k in ("n_gpus", "value", "ms_per_step", "frames_per_s_of_each_rank", "host_ms_per_step_of_each_rank", "distributed")}, open(rep, "w"), indent=1)
    # without the rehearsal hook a box with fewer devices than ranks is an error, not a silent 1-rank run
    env.pop("ADK_BENCH_ONE_GPU")
    if ranks == 2 and torch.cuda.device_count() < 8:
        r = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "8", "--steps", "2", "--warmup", "1"],
                           capture_output=True, text=True, timeout=300, env=env, cwd=root)
        assert r.returncode != 0 and "HIP device(s) visible" in r.stderr and not [l for l in r.stdout.splitlines() if l.startswith("{")]


@pytest.mark.parametrize("split16", [True, False], ids=["split16", "f32"])
def test_graph_replay_is_bit_identical_to_eager(gpu, ckpt_root, split16):
    """adk_program_set_graph: the captured launch sequences (one per cursor phase, period 12 at one frame per step) replay the
    same kernels with the same arguments, so every output bit equals the eager path -- across more than two periods, a
    reset_buffer + re-warm, a per-stream reset, and short steps in between (which run eagerly and leave the phase)."""
    hop, B, steps = HOP, 6, 40
    audio = np.stack([synth.synth_audio(321, s, (steps + 8) * hop) for s in range(B)])
    old = os.environ.get("ADK_GRAPH")
    try:
        os.environ["ADK_GRAPH"] = "0"
        ad_e = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 2, split16)
        os.environ["ADK_GRAPH"] = "1"
        ad_g = load_audiodec(ckpt_root, "vctk_v1", 1337, B, 2, split16)
    finally:
        if old is None:
            os.environ.pop("ADK_GRAPH", None)
        else:
            os.environ["ADK_GRAPH"] = old
    assert ad_g.tx_encoder.graph and ad_g.decoder.graph and not ad_e.decoder.graph
    assert ad_g.decoder._decoder().graph and ad_g.tx_encoder._encoder().graph          # the C++ side accepted the ring sizes

    def run(ad, x):
        idx = ad.tx_encoder.quantize(ad.tx_encoder.encode(x))
        return idx, ad.decoder.decode(ad.rx_encoder.lookup(idx))

    pos = 0
    side = torch.cuda.Stream(DEV)                                  # graphs are captured on a side stream (never on the legacy default one)
    torch.cuda.synchronize()                                       # the warm-ups above ran on the default stream
    with torch.no_grad(), torch.cuda.stream(side):
        for i in range(steps):
            frames = 1 if i in (17, 18, 19) else 2                 # three short steps: eager, and the phase is left for a while
            x = torch.from_numpy(audio[:, pos:pos + frames * hop])[:, None, :].to(DEV)
            pos += frames * hop
            if pos + 2 * hop > audio.shape[1]:
                pos = 0
            (ie, ye), (ig, yg) = run(ad_e, x), run(ad_g, x)
            assert torch.equal(ie, ig) and torch.equal(ye, yg), i
            if i == 25:                                            # whole-model reset + warm-up, then a per-stream reset
                for ad in (ad_e, ad_g):
                    ad.tx_encoder.reset_buffer(); ad.rx_encoder.reset_buffer(); ad.decoder.reset_buffer()
                    ad.tx_encoder.initial_encoder(8192, DEV)
                    ad.decoder.initial_decoder(ad.rx_encoder.initial_encoder(8192, DEV))
            if i == 30:
                for ad in (ad_e, ad_g):
                    ad.tx_encoder.reset_stream(2); ad.decoder.reset_stream(2)
    rep, cap, per = ad_g.decoder._decoder().graph_stats()
    assert per >= 1 and rep > 0 and 0 < cap <= per, (rep, cap, per)
    rep_e, cap_e, per_e = ad_g.tx_encoder._encoder().graph_stats()
    assert rep_e > 0 and cap_e <= per_e
    assert ad_e.decoder._decoder().graph_stats() == (0, 0, 0)
    from audiodec_amd import native
    assert native.device_flags() == 0


@pytest.mark.parametrize("model,B,max_frames", [("vctk_sym", 200, 1), ("vctk_sym", 100, 2), ("vctk_v1", 256, 1),
                                                ("vctk_sym", 1, 1), ("vctk_sym", 3, 2), ("vctk_v1", 24, 1)])
def test_fused_residual_units_are_bit_identical(gpu, ckpt_root, model, B, max_frames):
    """CausalResidualUnit.inference (residual_unit.py:78-81) as one launch (conv_rl16_kernel<..., FUSE>: K7 conv, act + split in
    registers -> LDS, 1x1 conv + residual) against the same unit as two launches (ADK_FUSE=0): every output bit equal, for the
    encoder's and the symmetric decoder's 32- / 64-channel blocks, 4- and 5-wave tilings; at few streams the time tiles are
    one n-tile long (ragged last tile at 300 steps)."""
    from audiodec_amd import program
    hop = HOP
    audio = np.stack([synth.synth_audio(77, s % 7, 3 * max_frames * hop) for s in range(B)])
    old, old_c = program.FUSE_RES_UNITS, program.FUSE_CHAINS
    try:
        program.FUSE_CHAINS = False                  # (whole chains as one launch: test_residual_chains_are_bit_identical)
        program.FUSE_RES_UNITS = True
        ad_f = load_audiodec(ckpt_root, model, 1337, B, max_frames, True)
        program.FUSE_RES_UNITS = False
        ad_u = load_audiodec(ckpt_root, model, 1337, B, max_frames, True)
    finally:
        program.FUSE_RES_UNITS, program.FUSE_CHAINS = old, old_c
    kf = [ad_f.tx_encoder._encoder().describe_op(i, max_frames) for i in range(ad_f.tx_encoder._encoder().n_ops)]
    ku = [ad_u.tx_encoder._encoder().describe_op(i, max_frames) for i in range(ad_u.tx_encoder._encoder().n_ops)]
    assert kf.count("conv_rl16_unit<32>") == 3 and kf.count("conv_rl16_unit<64>") == 3 and kf.count("(fused into the previous op)") == 6, kf
    assert not any("unit" in k or "fused" in k for k in ku), ku
    with torch.no_grad():
        for i in range(3):
            x = torch.from_numpy(audio[:, i * max_frames * hop:(i + 1) * max_frames * hop])[:, None, :].to(DEV)
            zf, zu = ad_f.tx_encoder.encode(x), ad_u.tx_encoder.encode(x)
            assert torch.equal(zf, zu), i
            idx = ad_f.tx_encoder.quantize(zf)
            yf = ad_f.decoder.decode(ad_f.rx_encoder.lookup(idx)); yu = ad_u.decoder.decode(ad_u.rx_encoder.lookup(idx))
            # (vctk_v1: conv_out + the last up-sampler as ONE launch runs on 16 x 16 x 32 MFMAs since round 6 -- the same products and chunk order as
            # the two-launch form on 32 x 32 x 16, another grouping of the f32 additions inside an instruction: f32 round-off, not bit-identity)
            if model == "vctk_v1":
                assert float((yf - yu).abs().max()) < 2e-6, (i, float((yf - yu).abs().max()))
            else:
                assert torch.equal(yf, yu), i
    from audiodec_amd import native
    assert native.device_flags() == 0


@pytest.mark.parametrize("model,B,max_frames,want", [
    ("vctk_v1", 256, 1, {"conv_rb16<32>": 2, "conv_rb16<64>": 2, "conv_rb16<128>": 2}),      # the benched size: 2 streams per workgroup at 128 channels
    ("vctk_v1", 3, 1, {"conv_rb16<32>": 2, "conv_rb16<64>": 2, "conv_rb16<128>": 2}),        # odd stream count: a workgroup with one stream
    ("vctk_sym", 37, 1, {"conv_rb16<32>": 2, "conv_rb16<64>": 2, "conv_rb16<128>": 2}),      # encoder + symmetric decoder: ELU units (K7 + 1x1, no bias)
    ("vctk_v0", 5, 1, {"conv_rb16<32>": 4, "conv_rb16<64>": 4, "conv_rb16<128>": 4}),        # MultiReceptiveField: K 3 / 7 / 11 blocks, one group
    ("vctk_v2", 9, 1, {"conv_rb16<32>": 2, "conv_rb16<64>": 2, "conv_rb16<128>": 2}),        # grouped K3 blocks
    ("vctk_v1", 4, 2, {"conv_rb16<128>": 2}),                                               # two frames per step: only the 128-channel chains still fit
])
def test_residual_chains_are_bit_identical(gpu, ckpt_root, model, B, max_frames, want):
    """A residual chain as ONE launch (csrc/conv_rb16.hip: HiFiGANResidualBlock.inference, residual_block.py:99-105; the three
    CausalResidualUnits of an encoder / decoder block, residual_unit.py:78-81) against the same ops launched one by one
    (ADK_CHAIN=0, ADK_FUSE=0) over 9 calls, full and SHORT steps mixed, with a reset_buffer + re-warm in between:
      * with the 32- / 64-channel chains fused (their per-op kernel is the rows-in-LDS kernel, same operation order) every bit
        of latent and waveform is equal -- including steps where the fused program falls back to per-op launches at run time
        (chain_max_channels = 0): the history a chain leaves in its intermediate rings, only the rows later calls read, is
        exactly what the per-op path leaves and expects;
      * with the 128-channel chains fused as well (their per-op kernel is the stream-K kernel, which splits K across
        workgroups: another association of the same sums) the results agree to f32 round-off."""
    from audiodec_amd import program, native
    hop = HOP
    old = program.FUSE_RES_UNITS, program.FUSE_CHAINS
    try:
        native.set_option("chain_min_blocks", 0)             # (the default since round 4; launches of <= 128 workgroups run the 8-wave variants)
        native.set_option("chain_max_channels", 64)          # the warm-up of the fused model runs 28 steps: keep it exact too
        program.FUSE_RES_UNITS, program.FUSE_CHAINS = True, True
        ad_f = load_audiodec(ckpt_root, model, 1337, B, max_frames, True)
        program.FUSE_RES_UNITS, program.FUSE_CHAINS = False, False
        ad_u = load_audiodec(ckpt_root, model, 1337, B, max_frames, True)
    finally:
        program.FUSE_RES_UNITS, program.FUSE_CHAINS = old
        native.set_option("chain_max_channels", 128)

    def progs(ad):
        dec = ad.decoder
        return [ad.tx_encoder._encoder()] + (list(dec._decoder_stages()) if hasattr(dec, "_decoder_stages") else [dec._decoder()])
    progs_f, progs_u = progs(ad_f), progs(ad_u)
    kf = [pr.describe_op(i, max_frames) for pr in progs_f for i in range(pr.n_ops)]
    ku = [pr.describe_op(i, max_frames) for pr in progs_u for i in range(pr.n_ops)]
    for name, n in want.items():
        assert kf.count(name) >= n, (name, kf)
    assert not any("rb16" in k or "fused" in k or "unit" in k for k in ku), ku
    # conv_out + the last up-sampler as one launch (conv_ou16, 16 x 16 x 32 MFMAs since round 6) against the two-launch form on 32 x 32 x 16: the
    # same products in the same chunk order, another grouping of the f32 additions inside an instruction -- the waveform agrees to f32 round-off
    # (a short step may fuse what a full one does not; conv_oc16 = the last conv_out + the output conv: its 1x1 conv sums in the order of
    # conv_sk16, the fused launch is nevertheless held to the same bound)
    ou16 = any(k in pr.describe_op(i, f) for k in ("conv_ou16", "conv_oc16") for pr in progs_f for i in range(pr.n_ops) for f in {1, max_frames})
    frames = [max_frames, max_frames, 1, max_frames, max_frames, 1, max_frames, max_frames, max_frames] if max_frames > 1 else [1] * 9
    maxc = [64, 64, 0, 64, 64, 64, 128, 128, 64]               # chain_max_channels of the fused model per call
    audio = np.stack([synth.synth_audio(91, s % 5, sum(frames) * hop) for s in range(B)])
    pos, exact = 0, True
    try:
        with torch.no_grad():
            for i, f in enumerate(frames):
                if i == 4:
                    native.set_option("chain_max_channels", 64)
                    for ad in (ad_f, ad_u):              # reset_buffer() + warm-up again (bin/stream.py:59-61, 68-76)
                        ad.tx_encoder.reset_buffer(); ad.decoder.reset_buffer()
                        ad.tx_encoder.initial_encoder(8192, DEV)
                        zq0 = ad.rx_encoder.initial_encoder(8192, DEV)
                        ad.decoder.initial_decoder(zq0)
                    exact = True
                x = torch.from_numpy(audio[:, pos:pos + f * hop])[:, None, :].to(DEV)
                pos += f * hop
                native.set_option("chain_max_channels", maxc[i])
                zf = ad_f.tx_encoder.encode(x)
                native.set_option("chain_max_channels", 0)
                zu = ad_u.tx_encoder.encode(x)
                exact = exact and maxc[i] <= 64              # once a 128-channel chain has run, the states differ by round-off
                if exact:
                    assert torch.equal(zf, zu), (i, float((zf - zu).abs().max()))
                else:
                    assert float((zf - zu).abs().max()) < 2e-5, (i, float((zf - zu).abs().max()))
                idx = ad_u.tx_encoder.quantize(zu)
                native.set_option("chain_max_channels", maxc[i])
                yf = ad_f.decoder.decode(ad_f.rx_encoder.lookup(idx))
                native.set_option("chain_max_channels", 0)
                yu = ad_u.decoder.decode(ad_u.rx_encoder.lookup(idx))
                if exact and not ou16:
                    assert torch.equal(yf, yu), (i, float((yf - yu).abs().max()))
                elif exact:
                    assert float((yf - yu).abs().max()) < 2e-6, (i, float((yf - yu).abs().max()))
                else:
                    assert float((yf - yu).abs().max()) < 2e-5, (i, float((yf - yu).abs().max()))
    finally:
        native.set_option("chain_max_channels", 128)
        native.set_option("chain_min_blocks", 0)
    assert native.device_flags() == 0


# ------------------------------------------------------------------------------------------------
# BASELINE configs 2 and 3 at EXACTLY their stream counts (kernel choice is a function of the stream count: few-streams time
# tiles, 8-wave chain kernels up to 128 workgroups), through the same checker bench.py's extra_configs runs beside the timing
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("split16", [True, False], ids=["split16", "f32"])
@pytest.mark.parametrize("cfg", ["cfg2_encoder_rvq_B32", "cfg3_full_B64"])
def test_survey_configs_2_and_3_match_oracle_at_their_stream_counts(gpu, ckpt_root, monkeypatch, cfg, split16):
    """`symAD_vctk_48000_hop300 encoder+RVQ only, batch=32` and `... full encode->RVQ->decode, batch=64 streaming (hop 300)`
    (BASELINE.json configs 2 / 3): 6 single-frame steps of a fresh `vctk_sym` model against the B-stream oracle -- latent and
    waveform <= 1e-4 max-abs, indices bit-exact (a flip only where the reference's own top-2 margin is below 1e-4)."""
    import bench
    monkeypatch.setenv("ADK_SPLIT16", "1" if split16 else "0")
    monkeypatch.setenv("ADK_VOCODER_STAGES", "1")
    B, decode = (32, False) if cfg.startswith("cfg2") else (64, True)
    res = bench.self_check(ckpt_root, DEV, B, 1, True, steps=6, model="vctk_sym", decode=decode)
    assert res["streams"] == B and res["steps"] == 6 and res["model"] == "vctk_sym"
    assert res["max_abs_dz"] < WAVE_TOL, res
    assert res["unexplained_flips"] == 0, res
    if decode:
        assert res["max_abs_dy"] < WAVE_TOL and res["streams_compared_to_the_end"] >= B - 2, res
    assert res["ok"], res
    from audiodec_amd import native
    assert native.device_flags() == 0


# ------------------------------------------------------------------------------------------------
# shadow rings (adk_op_desc.in_shadow / out_shadow): the producer's epilogue stores the split-f16 operand form once, the
# consumer stages it as it is -- the same values, so the outputs must not move by a bit
# ------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("model,B,max_frames,calls", [("vctk_v1", 256, 1, [1, 1, 1, 1, 1, 1]), ("vctk_sym", 33, 3, [1, 3, 2, 1, 3]), ("vctk_v0", 5, 2, [2, 1, 2])])
def test_shadow_rings_are_bit_identical(gpu, ckpt_root, monkeypatch, model, B, max_frames, calls):
    """Two models, one lowered with shadow rings and one without (ADK_SHADOW=0's switch), the same calls: ragged multi-frame steps
    (ring wrap-around), a per-stream reset to the warmed-up state in the middle, a full reset_buffer at the end.  Reference
    semantics are untouched (CausalConv1d.inference, layers/conv_layer.py:153-156: the state ring IS the pad_buffer; the shadow is a
    second encoding of the same rows): latent, indices and waveform are compared bit for bit."""
    from audiodec_amd import program, native
    seed = 1337
    monkeypatch.setattr(program, "SHADOW_RINGS", True)
    ad1 = load_audiodec(ckpt_root, model, seed, B, max_frames, True)
    progs = [ad1.tx_encoder._encoder()] + (list(ad1.decoder._decoder_stages()) if hasattr(ad1.decoder, "_decoder_stages") else [ad1.decoder._decoder()])
    n_in = sum(1 for pr in progs for i in range(pr.n_ops) if pr._ops[i].in_shadow)
    n_out = sum(1 for pr in progs for i in range(pr.n_ops) if pr._ops[i].out_shadow)
    assert n_in >= 7 and n_out >= 7, (n_in, n_out)              # the 256-channel blocks of encoder and decoder at least
    for pr in progs:
        for i in range(pr.n_ops):
            if pr._ops[i].out_shadow:
                assert pr.describe_op(i, 1).startswith(("conv_sk16<", "conv_gv16<")), (pr.op_names[i], pr.describe_op(i, 1))      # (few columns: the GEMV-shaped kernel, same epilogue)
    monkeypatch.setattr(program, "SHADOW_RINGS", False)
    ad0 = load_audiodec(ckpt_root, model, seed, B, max_frames, True)
    progs0 = [ad0.tx_encoder._encoder()] + (list(ad0.decoder._decoder_stages()) if hasattr(ad0.decoder, "_decoder_stages") else [ad0.decoder._decoder()])
    assert not any(pr._ops[i].in_shadow or pr._ops[i].out_shadow for pr in progs0 for i in range(pr.n_ops))
    assert sum(pr.n_rings for pr in progs) > sum(pr.n_rings for pr in progs0)
    total = sum(calls) * HOP
    audio = np.stack([synth.synth_audio(777, s, total) for s in range(B)])

    def run(ad):
        outs, pos = [], 0
        with torch.no_grad():
            for k, f in enumerate(calls):
                if k == len(calls) // 2:
                    for g_ in (ad.tx_encoder, ad.decoder):
                        g_.reset_stream(B - 1, warm=True)       # history rows of ring AND shadow rewritten from the captured state
                x = torch.from_numpy(audio[:, pos:pos + f * HOP].copy())[:, None, :].to(DEV)
                pos += f * HOP
                z = ad.tx_encoder.encode(x)
                idx = ad.tx_encoder.quantize(z)
                y = ad.decoder.decode(ad.rx_encoder.lookup(idx))
                outs.append((z.clone(), idx.clone(), y.clone()))
            ad.tx_encoder.reset_buffer(); ad.decoder.reset_buffer()
            x = torch.from_numpy(audio[:, :HOP].copy())[:, None, :].to(DEV)
            z = ad.tx_encoder.encode(x)
            outs.append((z.clone(), ad.tx_encoder.quantize(z).clone(), ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(z))).clone()))
        return outs

    o1, o0 = run(ad1), run(ad0)
    for k, ((z1, i1, y1), (z0, i0, y0)) in enumerate(zip(o1, o0)):
        assert torch.equal(z1, z0), f"call {k}: latent differs by {float((z1 - z0).abs().max()):.3e}"
        assert torch.equal(i1, i0), f"call {k}: indices differ"
        assert torch.equal(y1, y0), f"call {k}: waveform differs by {float((y1 - y0).abs().max()):.3e}"
    assert native.device_flags() == 0

