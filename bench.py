#!/usr/bin/env python3
"""bench.py -- the AudioDec streaming hot path on MI355X.

Workload at any N (weak scaling): per GPU 256 concurrent 48 kHz streams of the `vctk_v1` pipeline
(symAD encoder+projector -> 8-stage RVQ -> codebook lookup -> AudioDec-v1 HiFi-GAN vocoder), one
hop (300 samples = 1 frame) per stream per step, streaming state carried across steps
(BASELINE.json config 5 per-GPU share; 8 GPUs = its 2048 streams).  One step = one pass of the
hot path over one batch of synthetic audio already resident in HBM.

Prints ONE JSON line (rank 0).  value = frames/s for the whole job.
"""
import argparse
import json
import os
import sys
import tempfile
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MODEL = "vctk_v1"
SEED = 1337
HOP = 300
FP32_MFMA_PEAK_TFLOPS = 157.3      # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32 dense peak
F16_MFMA_PEAK_TFLOPS = 2500.0      # MI355X_MICROARCH.md: dense f16/bf16 MFMA peak; a split-f16 product sum issues 3 of them
HBM_PEAK_GBS = 8000.0              # MI355X_MICROARCH.md: HBM3E 8 TB/s spec
PMC_MARKER_N = 7654321                         # --pmc-markers: element count of the marker launches
PMC_TRAFFIC_FILE = "r6_pmc_traffic.csv"        # rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over this bench, pipelined schedule (tools/profile_round.sh)
STEADY_STATS_FILE = "r6_kernel_stats_steady.csv"   # rocprofv3 --kernel-trace over the timed steps of the same schedule (tools/trace_summary.py)
SERIAL_STATS_FILE = "r6_kernel_stats_serial.csv"   # ... of `bench.py --serial` (one HIP stream, nothing else on the chip)
PMC_TRAFFIC_SCRIPT = "tools/profile_round.sh"
# the same three captures of `bench.py --frames-per-step 5` (the reference streamer's default chunk): the secondary roofline of the named kernel
T5_STEADY_STATS_FILE, T5_SERIAL_STATS_FILE, T5_PMC_TRAFFIC_FILE = "r6_kernel_stats_T5_steady.csv", "r6_kernel_stats_T5_serial.csv", "r6_pmc_traffic_T5.csv"

def build_audiodec(root, device, streams, max_frames, model=None, guard=None):
    from audiodec_amd import synth
    from audiodec_amd.audiodec import AudioDec, assign_model
    if model is not None and model != MODEL:
        synth.write_model(root, model, SEED)        # (MODEL's checkpoints were written by main() from the broadcast state dicts)
    cwd = os.getcwd()
    os.chdir(root)
    try:
        sr, enc_ckpt, dec_ckpt = assign_model(model or MODEL)
        # guard=None (the headline): AudioDec's default -- on.  Direct calls check every program step before they return (one stream
        # synchronisation each); the pipeline object of the timed region defers the checks and repairs by replay
        # (audiodec_amd/pipeline.py).  guard=False: nothing is checked per step (the `unguarded` leg; what rounds 1-4 timed as `value`)
        ad = AudioDec(tx_device=device, rx_device=device, num_streams=streams, max_frames=max_frames, guard=guard)
        import contextlib, io
        with contextlib.redirect_stdout(io.StringIO()):
            ad.load_transmitter(enc_ckpt)
            ad.load_receiver(enc_ckpt, dec_ckpt)
    finally:
        os.chdir(cwd)
    return ad


class unguarded:
    """Context: the generators of `ad` with their guard off (the serial legs that price KERNELS or a bare device-complete latency take the
    per-step host synchronisations of the synchronous guard out; the guarded figure is reported beside them)."""

    def __init__(self, *ads):
        self.gens = [g for a in ads for g in (a.tx_encoder, a.rx_encoder, a.decoder) if g is not None]

    def __enter__(self):
        self.keep = [g.guard for g in self.gens]
        for g in self.gens:
            g.set_guard(False)

    def __exit__(self, *exc):
        for g, k in zip(self.gens, self.keep):
            g.set_guard(k)


def step(ad, x):
    z = ad.tx_encoder.encode(x)
    idx = ad.tx_encoder.quantize(z)
    zq = ad.rx_encoder.lookup(idx)
    return ad.decoder.decode(zq)


def TxRxPipeline(ad, dev, depth=4):
    """The schedule object of the timed region: audiodec_amd.pipeline.StreamingPipeline (the transmitter and every receiver program on
    their own HIP streams, batches handed over by events, the guard deferred), with this file's tuning knobs.

    ADK_BENCH_PRIO (tuning): HIP stream priorities of the transmitter, the receiver and the further vocoder stages (0 / -1).
    Measured, 100 steps, three runs each on one box: all equal 249.0-249.4 k frames/s, transmitter high 250.4-253.1 k, no
    difference over 20 steps; round 4, final build, same box, alternating: equal priorities 283.6 / 288.7 k, transmitter high
    287.7 / 288.4 k: no difference, equal priorities stay.
    ADK_BENCH_RVQ (tuning): the HIP stream the residual-VQ search of a batch is launched on -- tx (with the encoder, as the
    reference's transmitter thread does; default), rx, last, own (a fourth stream).  Measured, two alternating rounds on one box
    (tools/ab_multi.sh rq ADK_BENCH_RVQ 2 tx rx last own): tx 285.4 / 285.7 k frames/s, rx 285.2 / 281.6 k, last 244.4 / 243.8 k,
    own 244.3 / 244.2 k -- a fourth stream shares one of the runtime's 4 hardware queues (profiles/r4_few_streams.md section 5).
    ADK_BENCH_WORKGROUPS (tuning): cap on the persistent workgroups of a stream-K launch (round 1: 256 best; since the tile-aligned
    ranges of round 2 the library default is as fast: tools/run_r3b.sh)."""
    from audiodec_amd.pipeline import StreamingPipeline
    prio = [int(v) for v in os.environ.get("ADK_BENCH_PRIO", "0,0,0,0").split(",")] + [0, 0, 0, 0]
    pipe = StreamingPipeline(ad, dev, depth=depth, priorities=prio, rvq_stream=os.environ.get("ADK_BENCH_RVQ", "tx"))
    wg = int(os.environ.get("ADK_BENCH_WORKGROUPS", "0"))
    if wg > 0 and pipe.n_dec >= 2 and getattr(ad.decoder, "split16", False):
        ad.tx_encoder.set_workgroups(wg)
        ad.decoder.set_workgroups(wg)
    return pipe


def _programs_of(ad):
    progs = {"encoder": ad.tx_encoder._encoder()}
    stages = ad.decoder._decoder_stages() if hasattr(ad.decoder, "_decoder_stages") else [ad.decoder._decoder()]
    for i, pr in enumerate(stages):
        progs["decoder" if i == 0 else f"decoder{i}"] = pr
    return progs


def op_profile(ad, xs, streams, n_steps, fps=1, pipe=None, burst=7, at=3):
    """Per-op HIP-event durations (events recorded on the launch stream by the C++ runner around every op).

    pipe=None: the SERIAL schedule -- one HIP stream, one batch at a time, nothing else on the chip.
    pipe=TxRxPipeline: the schedule `value` is timed in -- n_steps bursts of `burst` pipeline steps, of which step `at` (batches before
    AND behind it in flight on the other HIP streams) records the events; a program's events are re-recorded by every profiled step,
    so only one step per burst can be read, and reading needs a device synchronisation, which is why this is a region of its own
    behind the timed one and not the timed region itself."""
    progs = _programs_of(ad)
    acc = {k: np.zeros(p.n_ops) for k, p in progs.items()}
    if pipe is None:
        for p in progs.values():
            p.set_profiling(True)
        for i in range(n_steps):
            step(ad, xs[i % len(xs)])
            for k, p in progs.items():
                acc[k] += np.asarray(p.last_op_ms())
        for p in progs.values():
            p.set_profiling(False)
    else:
        for i in range(n_steps):
            torch.cuda.synchronize()
            pipe.enter()
            for j in range(burst):
                if j == at:
                    for p in progs.values():
                        p.set_profiling(True)
                pipe.step(xs[(i * burst + j) % len(xs)])
                if j == at:
                    for p in progs.values():
                        p.set_profiling(False)          # (a flag of the host-side runner: the recorded events stay as they are)
            pipe.exit()
            torch.cuda.synchronize()
            for k, p in progs.items():
                p.set_profiling(True)
                acc[k] += np.asarray(p.last_op_ms())
                p.set_profiling(False)
    rows = []
    for k, p in progs.items():
        for i in range(p.n_ops):
            op = p._ops[i]
            flops = 0.0
            if op.kind == 0:
                c = op.conv
                flops = 2.0 * c.groups * c.cout_g * c.taps * c.cin_g * op.rate_out * streams * fps
            nbytes = 0.0
            if op.kind == 0:
                # compulsory bytes of the launch: input rows incl. history (once), weights (once), outputs, residual
                c = op.conv
                cin_tot = c.cin_g * (c.groups if c.in_group_stride else 1)
                t_out = op.rate_out * fps
                nbytes = 4.0 * (streams * (t_out * c.stride + c.hist) * cin_tot + c.groups * c.cout_g * c.taps * c.cin_g
                                + streams * t_out * c.groups * c.cout_g * (2 if op.res_ring >= 0 else 1))
            rows.append(dict(prog=k, name=p.op_names[i], kernel=p.describe_op(i, fps), ms=acc[k][i] / n_steps, flops=flops, bytes=nbytes, op=op))
    return rows


def _rocprof_name(dom):
    """bench / describe_op kernel name -> substring(s) of the demangled kernel name in the rocprofv3 CSVs."""
    import re
    m = re.match(r"conv_sk(16)?<(\d+)x(\d+)>", dom)
    if m:
        cfg = {"64x64": "<2, 2, 1,", "128x64": "<4, 1, 2,", "32x128": "<1, 4, 1,"}.get(f"{m.group(2)}x{m.group(3)}")
        return ("conv_sk_kernel" + cfg, "true" if m.group(1) else "false") if cfg else None
    m = re.match(r"conv_rb16<(\d+)>", dom)
    if m:
        return (f"conv_rb16_kernel<{m.group(1)},",)
    m = re.match(r"conv_rl16(_unit)?<(\d+)>", dom)
    if m:
        return (f"conv_rl16_kernel<{m.group(2)},",)
    m = re.match(r"conv_up16<(\d+)>", dom)
    if m:
        return ("conv_up16_kernel<",)
    if dom.startswith("conv_ou16<"):
        return ("conv_ou16_",)                       # conv_ou16_w8_kernel<ACT, MT2> (round 6; conv_ou16_dma_kernel / conv_ou16_kernel before)
    return None


def _profile_csv(name):
    """(rows, header comments) of a committed profile CSV, or (None, {})."""
    import csv
    path = os.path.join(ROOT, "profiles", name)
    if not os.path.exists(path):
        return None, {}
    lines = open(path).read().splitlines()
    meta = {}
    for l in lines:
        if l.startswith("# ") and ":" in l:
            k, v = l[2:].split(":", 1)
            meta[k.strip()] = v.strip()
    return list(csv.DictReader(l for l in lines if not l.startswith("#"))), meta


def _digest_now():
    import __graft_entry__ as g
    return g.kernel_source_digest()[:16]


BENCH_CONFIG = None        # set by main(): what tools/profile_round.sh records as `# bench_config:` in the captures it writes


def bench_config_string(streams, stages, fps, precision, guard, serial=False):
    return f"streams={streams} stages={stages} frames_per_step={fps} precision={precision} guard={guard} rvq={os.environ.get('ADK_BENCH_RVQ', 'tx')}" + (" serial" if serial else "")


def _capture_stale(meta, config=None):
    """A committed capture is stale unless it was taken from THIS build of the kernels, THIS host-side lowering / schedule (schedule_digest)
    and the bench configuration of this run (captures from before round 5 carry no schedule digest: stale by definition once it matters)."""
    import __graft_entry__ as g
    if meta.get("source_digest", "").split()[0:1] != [_digest_now()]:
        return True
    if meta.get("schedule_digest", "").split()[0:1] != [g.schedule_digest()[:16]]:
        return True
    cfg = meta.get("bench_config", "unknown")
    want = (config or BENCH_CONFIG or "").replace(" serial", "")
    return cfg.replace(" serial", "") != want


def pmc_traffic(dom, traffic_file=None, config=None):
    """HBM-side bytes per launch of kernel `dom` from the committed rocprofv3 PMC passes (FETCH_SIZE doubled as MI355X_MICROARCH.md
    prescribes for 16 B/lane streaming reads, + WRITE_SIZE), launch-weighted mean over ALL its launches between the two
    --pmc-markers of those passes (= the timed steps of this same bench in the pipelined schedule; tools/pmc_summary.py).  PMC counters
    cannot be collected from inside this process.  Returns (bytes or None, stale): stale = the kernels were rebuilt from other
    sources since the capture (the file carries the source digest of its build)."""
    rows, meta = _profile_csv(traffic_file or PMC_TRAFFIC_FILE)
    sub = _rocprof_name(dom)
    if rows is None or sub is None:
        return None, None
    num = den = 0.0
    for r in rows:
        if all(t in r["kernel"] for t in sub):
            n = float(r["launches"])
            num += n * (float(r["FETCH_KB_x2_corrected"]) + float(r["WRITE_SIZE_KB_avg"])) * 1024.0
            den += n
    stale = _capture_stale(meta, config)
    return (round(num / den) if den else None), stale


def rocprof_duration(dom, stats_file=None, config=None):
    """Average duration (us) of kernel `dom` over the timed steps of the bench, from a committed rocprofv3 kernel trace (profiles/
    STEADY_STATS_FILE: the pipelined schedule, SERIAL_STATS_FILE: `--serial`; the dispatch's own start/end, no launch gap, no event
    record), and whether that capture is stale."""
    rows, meta = _profile_csv(stats_file or STEADY_STATS_FILE)
    sub = _rocprof_name(dom)
    if rows is None or sub is None:
        return None, None
    num = den = 0.0
    for r in rows:
        if all(t in r["kernel"] for t in sub):
            num += float(r["total_us"]); den += float(r["launches"])
    stale = _capture_stale(meta, config)
    return (round(num / den, 2) if den else None), stale


FUSED = "(fused into the previous op)"
IN_RING_WRITE = "(in the launch of the ring write)"      # the Cin = 1 conv behind a ring write: one launch (conv_cin1w_kernel)


def launches_of(rows, streams, fps=1):
    """Per-op rows -> per-LAUNCH records.  An op the runner folded into its predecessor's launch (a residual unit or a whole
    residual chain run as one kernel) adds its flops to that launch.  Time: the runner records one event behind every op, so a launch
    that covers n ops is followed by n events back to back; the kernel sits between the event in front of the head op and the one
    behind it -- `ms` = the HEAD op's event time (kernel + ONE event record); the n - 1 further "durations" are event records with
    no kernel in between, kept as `ms_events_only` (what an event record costs on this stream, measured in the run).
    Algorithmic bytes of a fused launch: chain input (new rows + history) + every conv's weights + the history rows the later convs
    read from / leave in their state rings + the chain output, once each -- what no implementation of the streaming recurrence can
    avoid moving."""
    out = []
    for r in rows:
        if r["kernel"] in (FUSED, IN_RING_WRITE) and out and out[-1]["prog"] == r["prog"]:
            L = out[-1]
            L["ms_events_only"] += r["ms"]; L["flops"] += r["flops"]; L["ops"].append(r)
            continue
        out.append(dict(prog=r["prog"], name=r["name"], kernel=r["kernel"], ms=r["ms"], ms_events_only=0.0, flops=r["flops"], bytes=r["bytes"], ops=[r], op=r["op"]))
    for L in out:
        if len(L["ops"]) > 1 and L["ops"][0]["op"].kind != 0:
            L["bytes"] = sum(q["bytes"] for q in L["ops"][1:])            # ring write + conv: the conv's bytes (the ring rows are written once either way)
        elif len(L["ops"]) > 1:
            ops = [q["op"] for q in L["ops"]]
            c0 = ops[0].conv
            t = ops[0].rate_out * fps
            m = c0.groups * c0.cout_g
            b = streams * (t + c0.hist) * c0.cin_g * (c0.groups if c0.in_group_stride else 1)             # chain input
            b += sum(o.conv.groups * o.conv.cout_g * o.conv.taps * o.conv.cin_g for o in ops)             # weights
            b += sum(2 * streams * min(o.conv.hist, 10 ** 9) * m for o in ops[1:])                         # state rows read + written back
            b += streams * t * m                                                                           # chain output
            L["bytes"] = 4.0 * b
    return out


def mean_launch_bytes(launches, dom):
    """Mean compulsory bytes per launch over the launches of kernel `dom` in one step (the launches pmc_traffic averages over)."""
    sel = [L["bytes"] for L in launches if L["kernel"] == dom and L["op"].kind == 0]
    return round(sum(sel) / len(sel)) if sel else None


def event_record_ms(rows):
    """What ONE event record costs on the launch stream: median "duration" of the ops that ran inside another op's launch (two event
    records with no kernel in between).  0 when the step has no such op."""
    gaps = [r["ms"] for r in rows if r["kernel"] in (FUSED, IN_RING_WRITE)]
    return float(np.median(gaps)) if gaps else 0.0


def kernel_table(rows, streams, fps=1):
    """{kernel: {ms, ms_est, flops, launches, bytes}} over one step.  ms = the raw HIP-event time of its launches (each includes ONE event
    record); ms_est = the same minus ONE measured event record per launch (an ESTIMATE of the kernels alone, floored at half the raw)."""
    launches = launches_of(rows, streams, fps)
    ev = event_record_ms(rows)
    by = {}
    for L in launches:
        d = by.setdefault(L["kernel"], dict(ms=0.0, ms_est=0.0, flops=0.0, launches=0, bytes=0.0))
        d["ms"] += L["ms"]; d["ms_est"] += max(L["ms"] - ev, 0.5 * L["ms"]); d["flops"] += L["flops"]; d["launches"] += 1; d["bytes"] += L.get("bytes", 0.0)
    return launches, by, ev


def _frac(flops, ms, peak):
    return round(flops / (ms * 1e-3) / 1e12 / peak, 4) if ms and ms > 0 else None


def roofline_from(rows_serial, streams, fps=1, split16=False, rows_pipe=None):
    """The roofline objects of the JSON line.  rows_serial: per-op events of the serial schedule; rows_pipe: of the pipelined schedule
    (None for --serial runs).  The dominant kernel is picked by its share of the event time in the schedule `value` is timed in; `frac`
    prices it on its dispatch duration in THAT schedule (committed rocprofv3 kernel trace of this build; live events when there is none),
    with the other schedule, the live event figures and the minus-one-event-record estimate beside it."""
    launches_s, by_s, ev_s = kernel_table(rows_serial, streams, fps)
    if rows_pipe is not None:
        launches_p, by_p, ev_p = kernel_table(rows_pipe, streams, fps)
    else:
        launches_p, by_p, ev_p = launches_s, by_s, ev_s
    dom = max(by_p, key=lambda k: by_p[k]["ms"])
    d, ds = by_p[dom], by_s[dom]
    # algorithmic (f32-equivalent) flops against the matrix-core peak of the instruction the kernel issues: the exact-f32
    # MFMA, or -- for the split-f16 kernels -- the dense f16 MFMA peak divided by the 3 instructions per product sum
    peak = F16_MFMA_PEAK_TFLOPS / 3.0 if (split16 and "16" in dom.split("<")[0]) else FP32_MFMA_PEAK_TFLOPS
    traffic, stale = pmc_traffic(dom)
    rp_us, rp_stale = rocprof_duration(dom, STEADY_STATS_FILE)
    rs_us, rs_stale = rocprof_duration(dom, SERIAL_STATS_FILE)
    fpl = d["flops"] / d["launches"]
    sched = "pipelined" if rows_pipe is not None else "serial"
    # `frac`: the dominant kernel's algorithmic flops over its DISPATCH duration (start to end of execution) in the schedule `value` is
    # timed in.  A HIP event pair on a stream that shares the chip with two other streams also times the wait for free compute units in
    # front of the kernel (conv_sk16: 54 us between its events, 37 us of dispatch), so the dispatch duration comes from the rocprofv3 kernel
    # trace of the same command committed under profiles/ -- when that trace is of THIS build (source digest); otherwise, and always beside
    # it, the live event figures.
    rp_fresh = rp_us is not None and rp_stale is False and sched == "pipelined"
    rs_fresh = rs_us is not None and rs_stale is False and sched == "serial"
    if rp_fresh or rs_fresh:
        us = rp_us if rp_fresh else rs_us
        achieved = fpl / (us * 1e-6) / 1e12
        src = f"rocprofv3 kernel trace of the timed steps, {sched} schedule (profiles/{STEADY_STATS_FILE if rp_fresh else SERIAL_STATS_FILE}, same source digest as this build)"
    else:
        achieved = d["flops"] / (d["ms"] * 1e-3) / 1e12 if d["ms"] > 0 else 0.0
        src = f"live HIP events, {sched} schedule (no committed rocprofv3 trace of this build: the kernels were rebuilt since profiles/{STEADY_STATS_FILE})"
    roof = {"kernel": dom, "bound": "mfma", "achieved": round(achieved, 2), "peak": round(peak, 1),
            "unit": "TFLOP/s", "frac": round(achieved / peak, 4), "frac_schedule": sched, "frac_source": src,
            "frac_events_pipelined": _frac(d["flops"], d["ms"], peak) if rows_pipe is not None else None,
            "frac_events_serial": _frac(ds["flops"], ds["ms"], peak),
            "frac_events_serial_minus_event_estimate": _frac(ds["flops"], ds["ms_est"], peak),
            "frac_pipelined": _frac(fpl, rp_us * 1e-3, peak) if rp_us else None,
            "frac_serial": _frac(fpl, rs_us * 1e-3, peak) if rs_us else None,
            "traffic": traffic, "traffic_stale": stale,
            "algorithmic_bytes_per_launch": mean_launch_bytes(launches_p, dom),
            "traffic_note": "both are means per launch over all launches of this kernel in the timed steps: traffic = FETCH_SIZE x2 + "
                            f"WRITE_SIZE from the committed PMC passes over this bench in the SAME pipelined schedule (profiles/{PMC_TRAFFIC_FILE}, "
                            f"{PMC_TRAFFIC_SCRIPT}, steady-state launches picked out by --pmc-markers; PMC counters cannot be read from inside the "
                            "bench process), not collected live -- traffic_stale says whether the kernels were rebuilt from other sources since; "
                            "algorithmic = input rows incl. history + weights + outputs (+ residual / state rows), once each",
            "launches_per_step": d["launches"],
            "avg_launch_us": rp_us if rp_fresh else (rs_us if rs_fresh else round(1e3 * d["ms"] / d["launches"], 2)),
            "avg_launch_us_pipelined": rp_us, "avg_launch_us_serial": rs_us,
            "avg_launch_us_events_pipelined": round(1e3 * d["ms"] / d["launches"], 2) if rows_pipe is not None else None,
            "avg_launch_us_events_serial": round(1e3 * ds["ms"] / ds["launches"], 2),
            "event_record_us": round(1e3 * ev_p, 2), "event_record_us_serial": round(1e3 * ev_s, 2),
            "rocprof_stale": None if rp_stale is None and rs_stale is None else bool(rp_stale or rs_stale),
            "duration_note": "frac / frac_pipelined / frac_serial = flops_per_launch / avg_launch_us_{pipelined, serial} / peak: the launch-weighted mean "
                             f"dispatch duration of this kernel over the timed steps in the committed rocprofv3 kernel traces (profiles/{STEADY_STATS_FILE}: "
                             f"the three-stream schedule `value` is timed in; profiles/{SERIAL_STATS_FILE}: --serial; tools/profile_round.sh, "
                             "tools/trace_summary.py) -- reproducible from those files alone; rocprof_stale = the kernels were rebuilt since.  *_events_*: live HIP "
                             "events of THIS run on the launch stream (the C++ runner records one behind every op; every figure includes ONE event record, "
                             "event_record_us, measured on the ops that ran inside another op's launch): serial they agree with the dispatch durations to "
                             "the cost of the event; pipelined they also contain the wait for compute units the other two streams hold.  "
                             "*_minus_event_estimate: one measured event record subtracted per launch -- an estimate",
            "flops_per_launch": fpl, "share_of_step_kernel_time": round(d["ms"] / sum(v["ms"] for v in by_p.values()), 3),
            "launches_per_step_all_kernels": len(launches_p)}
    # the north-star's named kernel: fused LeakyReLU -> ConvTranspose1d(64->32, s3) + bias (last upsampler) -- since round 3 the launch
    # that contains it also runs the 1x1 conv_out (192 -> 64) in front of it (conv_ou16): bytes and time are those of THAT launch
    roof_ct = convtr_roofline(launches_s, launches_p if rows_pipe is not None else None, ev_s, ev_p, streams, fps)
    kernels = {k: {"ms_per_step": round(v["ms"], 4), "ms_per_step_serial": round(by_s[k]["ms"], 4) if k in by_s else None, "launches": v["launches"],
                   "tflops": round(v["flops"] / (v["ms"] * 1e-3) / 1e12, 2) if v["ms"] > 0 else 0.0} for k, v in by_p.items()}
    return roof, roof_ct, kernels


def convtr_roofline(launches_s, launches_p, ev_s, ev_p, streams, fps):
    """HBM roofline of the launch that contains upsamples.3 (the north-star's named kernel), from per-launch event records."""
    ct = [L for L in launches_s if any(q["name"] == "upsamples.3" for q in L["ops"])]
    if not ct:
        return None
    L = ct[0]
    up = [q for q in L["ops"] if q["name"] == "upsamples.3"][0]
    c = up["op"].conv
    t_in = up["op"].rate_out * fps
    cin, cout, s = c.cin_g, c.cout_real, c.up
    fused = len(L["ops"]) > 1
    if fused:
        c1 = L["ops"][0]["op"].conv                      # the 1x1 conv: reads cin1 channels per step, the 64-channel tensor stays on chip
        bytes_alg = 4.0 * (c1.cin_g * t_in + cin + cout * t_in * s) * streams + 4.0 * (c1.cin_g * c1.cout_g + cin * cout * 2 * s)
        what = f"{L['kernel']} blocks.2.conv_out (1x1 {c1.cin_g}->{c1.cout_g}) + LeakyReLU + upsamples.3 (ConvTranspose1d {cin}->{cout} s{s} + bias), one launch"
    else:
        bytes_alg = 4.0 * (cin * (t_in + 1) + cout * t_in * s) * streams + 4.0 * cin * cout * 2 * s
        what = L["kernel"] + " upsamples.3 (LeakyReLU+ConvTranspose1d 64->32 s3 +bias)"
    gbs = lambda ms: round(bytes_alg / (ms * 1e-3) / 1e9, 1) if ms and ms > 0 else None
    fr = lambda ms: round(bytes_alg / (ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4) if ms and ms > 0 else None
    ms_s = L["ms"]
    ms_p = None
    if launches_p is not None:
        lp = [q for q in launches_p if any(o["name"] == "upsamples.3" for o in q["ops"])]
        ms_p = lp[0]["ms"] if lp else None
    sched = "pipelined" if ms_p is not None else "serial"
    rp_us, rp_stale = rocprof_duration(L["kernel"], STEADY_STATS_FILE)
    rs_us, rs_stale = rocprof_duration(L["kernel"], SERIAL_STATS_FILE)
    rp_fresh = rp_us is not None and rp_stale is False and sched == "pipelined"
    rs_fresh = rs_us is not None and rs_stale is False and sched == "serial"
    ms_ev = ms_p if ms_p is not None else ms_s
    ms_main = 1e-3 * rp_us if rp_fresh else (1e-3 * rs_us if rs_fresh else ms_ev)      # as roofline.frac: dispatch duration of the timed schedule when a trace of this build is committed
    return {"kernel": what, "bound": "hbm", "achieved": gbs(ms_main), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": fr(ms_main),
            "frac_schedule": sched,
            "frac_source": ("rocprofv3 kernel trace of the timed steps (profiles/, same source digest as this build)" if (rp_fresh or rs_fresh)
                            else "live HIP events (no committed rocprofv3 trace of this build)"),
            "frac_pipelined": fr(rp_us * 1e-3) if rp_us else None, "frac_serial": fr(rs_us * 1e-3) if rs_us else None,
            "frac_events_pipelined": fr(ms_p), "frac_events_serial": fr(ms_s),
            "frac_events_serial_minus_event_estimate": fr(max(ms_s - ev_s, 0.5 * ms_s)) if 0.0 < ev_s < 0.015 else None,
            "traffic": pmc_traffic(L["kernel"])[0], "avg_launch_us": round(1e3 * ms_main, 2),
            "avg_launch_us_pipelined": rp_us, "avg_launch_us_serial": rs_us,
            "avg_launch_us_events_pipelined": round(1e3 * ms_p, 2) if ms_p is not None else None, "avg_launch_us_events_serial": round(1e3 * ms_s, 2),
            "event_record_us_serial": round(1e3 * ev_s, 2),
            "rocprof_stale": None if rp_stale is None and rs_stale is None else bool(rp_stale or rs_stale),
            "bytes_per_launch": bytes_alg, "fused_with_conv_out": fused,
            "fp32_tflops": round(L["flops"] / (ms_main * 1e-3) / 1e12, 2),
            "how": "frac / frac_pipelined / frac_serial: bytes_per_launch over the launch's dispatch duration in the committed rocprofv3 kernel traces of the "
                   "timed steps (three-stream schedule / --serial); *_events_*: live HIP events around the launch in this run (each includes one event "
                   "record; pipelined also the wait for compute units held by the other streams); *_minus_event_estimate subtracts one measured event record"}


def convtr_standalone(dev, sd_dec, B, fps, split16, iters=300):
    """The north-star's named kernel by itself: fused LeakyReLU(0.1) -> ConvTranspose1d(64 -> 32, K 6, stride 3) + bias with the
    vocoder's own (weight-norm folded) weights, `iters` back-to-back launches on the current HIP stream bracketed by HIP events
    -- the per-op events of the pipeline profile also time the gap to the previous launch and the event itself (~4 us)."""
    from audiodec_amd import layers, native
    w = torch._weight_norm(sd_dec["upsamples.3.deconv.weight_v"].float(), sd_dec["upsamples.3.deconv.weight_g"].float(), 0)
    bias = sd_dec["upsamples.3.deconv.bias"].float()
    t_in = 100 * fps
    m = layers.CausalConvTranspose1d(64, 32, 6, 3, device=dev, batch=B, max_len=t_in).load(w, bias)
    m.set_activation("LeakyReLU", 0.1)
    m.impl = native.IMPL_SPLIT16 if split16 else native.IMPL_AUTO
    g = torch.Generator().manual_seed(SEED)
    m.inference(torch.randn(B, 64, t_in, generator=g))
    torch.cuda.synchronize()
    us = m.time_kernel(t_in, iters)            # the launch loop and its HIP events run inside the library (no Python per launch)
    return us, m.last_kernel


def convtr_t5(root, dev, sd_dec, B, split16, fps=5, check=True):
    """Secondary roofline of the north-star's named kernel at the reference streamer's DEFAULT chunk: demoStream.py:28 frame_size = 1500
    samples = 5 hops per call (T = 5), where one launch of upsamples.3 moves 5x the rows of the headline's single-frame step.  A second
    model (max_frames = 5) is profiled per op on one HIP stream; the fused conv_out + upsamples.3 launch takes at most 128 input steps per
    stream, so at 500 the transposed conv runs as its own streaming launch (conv_up16), which is what this line prices."""
    ad5 = build_audiodec(root, dev, B, fps)
    from audiodec_amd import synth
    xs5 = [torch.from_numpy(np.stack([synth.synth_audio(SEED + 500 + j, s, HOP * fps) for s in range(B)]))[:, None, :].to(dev) for j in range(2)]
    for i in range(4):
        step(ad5, xs5[i % 2])
    torch.cuda.synchronize()
    with unguarded(ad5):
        rows5 = op_profile(ad5, xs5, B, 6, fps)
    launches, by, ev = kernel_table(rows5, B, fps)
    res = convtr_roofline(launches, None, ev, ev, B, fps) or {}
    for k in ("frac_pipelined", "frac_serial", "frac_events_pipelined", "avg_launch_us_pipelined", "avg_launch_us_serial", "avg_launch_us_events_pipelined",
              "rocprof_stale", "traffic"):
        res.pop(k, None)                   # (the committed traces / PMC passes are of the single-frame headline schedule)
    # rocprofv3 evidence of THIS configuration: kernel traces (three-stream schedule and --serial) and PMC passes of
    # `bench.py --frames-per-step 5 --pmc-markers` (tools/profile_round.sh, section T5), fresh when taken from this build / lowering
    cfg5 = bench_config_string(B, os.environ.get("ADK_VOCODER_STAGES", "2"), fps, "split16" if split16 else "f32", "default")
    kern5 = "conv_up16<64>"
    bpl = res.get("bytes_per_launch")
    rp_us, rp_stale = rocprof_duration(kern5, T5_STEADY_STATS_FILE, cfg5)
    rs_us, rs_stale = rocprof_duration(kern5, T5_SERIAL_STATS_FILE, cfg5)
    tr5, tr5_stale = pmc_traffic(kern5, T5_PMC_TRAFFIC_FILE, cfg5)
    if bpl:
        hb = lambda us_: round(bpl / (us_ * 1e-6) / 1e9 / HBM_PEAK_GBS, 4) if us_ else None
        res.update({"frac_pipelined": hb(rp_us), "frac_serial": hb(rs_us), "avg_launch_us_pipelined": rp_us, "avg_launch_us_serial": rs_us,
                    "traffic": tr5, "traffic_stale": tr5_stale, "rocprof_stale": None if rp_stale is None and rs_stale is None else bool(rp_stale or rs_stale)})
        if rp_us and rp_stale is False:
            res.update({"frac": hb(rp_us), "achieved": round(bpl / (rp_us * 1e-6) / 1e9, 1), "avg_launch_us": rp_us, "frac_schedule": "pipelined",
                        "frac_source": f"rocprofv3 kernel trace of the timed steps of bench.py --frames-per-step {fps}, three-stream schedule "
                                       f"(profiles/{T5_STEADY_STATS_FILE}, same source / schedule digest as this build)"})
    us, kname = convtr_standalone(dev, sd_dec, B, fps, split16)
    b_alone = 4.0 * (64 * (100 * fps + 1) + 32 * 300 * fps) * B + 4.0 * 64 * 32 * 6
    res["transposed_conv_alone"] = {"kernel": kname, "avg_launch_us": round(us, 2), "bytes_per_launch": b_alone,
                                    "achieved": round(b_alone / (us * 1e-6) / 1e9, 1), "frac": round(b_alone / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                    "how": f"300 back-to-back launches of upsamples.3 alone, {B} streams x {100 * fps} input steps, HIP events inside the library"}
    pipe5 = TxRxPipeline(ad5, dev)
    n = 20
    pipe5.enter()
    for i in range(6):
        pipe5.step(xs5[i % 2])
    pipe5.exit()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    pipe5.enter()
    for i in range(n):
        pipe5.step(xs5[i % 2])
    pipe5.exit()
    torch.cuda.synchronize()
    e = time.perf_counter() - t0
    res["frames_per_step_per_stream"] = fps
    res["pipeline_frames_per_s"] = round(B * fps * n / e, 1)
    res["pipeline_ms_per_step"] = round(1e3 * e / n, 4)
    if check:
        # this configuration selects other kernels than the headline (conv_up16 instead of conv_ou16, longer time tiles in the chains): it is
        # checked against the CPU oracle in its own right -- fresh model, same stream count, same schedule object, 2 steps of 5 frames
        c = self_check(root, dev, B, fps, False, steps=2)
        res["self_check"] = {k: c[k] for k in ("ok", "streams", "steps", "max_abs_dz", "max_abs_dy", "indices_equal", "frames_with_flipped_indices",
                                               "unexplained_flips", "rvq_decisions", "min_reference_top2_margin", "oracle_cpu_s")}
        res["self_check"]["frames_per_step_per_stream"] = fps
    res["what"] = (f"the same workload at {fps} frames per stream per call (demoStream.py:28: frame_size 1500 = 5 hops, the reference streamer's default "
                   "chunk); serial per-op HIP events + the kernel alone; NOT the headline configuration (1 frame per call)")
    return res


def cpu_baseline(budget_s=12.0, threads=4, stack_calls=0):
    """The CPU port (oracle = the reference's own ATen CPU kernels, minus its inspect.stack() cost)
    streaming ONE stream of the same pipeline frame by frame on the host cores.

    stack_calls > 0 adds that many `inspect.stack()` calls per frame: the reference evaluates one per check_mode call
    site it passes (models/utils.py:13-15 from encoder.py:77,138 and projector.py:53 -- 1 + 4 + 1 = 6 per encoded
    frame for this model; the HiFi-GAN vocoder has none), SURVEY.md 3.4.  The reference itself cannot travel to the GPU
    box, so its "as is" cost is EMULATED this way and labelled as such."""
    import inspect
    from audiodec_amd import synth, configs
    from oracle import audiodec_oracle as O
    torch.set_num_threads(threads)
    _, enc_tag, _, dec_tag, _ = configs.alias(MODEL)
    mt_d, _, pd = configs.experiment(dec_tag)
    _, _, pe = configs.experiment(enc_tag)
    tx = O.AutoEncoderOracle(synth.synth_state_dict(enc_tag, SEED), pe, 1)
    zq0 = tx.initial_encoder(8192)
    dec = O.build_decoder_oracle(synth.synth_state_dict(dec_tag, SEED), mt_d, pd, 1)
    dec.initial_decoder(zq0)
    x = torch.from_numpy(synth.synth_audio(SEED, 0, 64 * HOP))[None, None, :]
    n, t0 = 0, None
    with torch.no_grad():
        for i in range(100000):
            if i == 3:
                t0 = time.perf_counter()           # 3 warm-up frames
            f = i % 64
            for _ in range(stack_calls):
                inspect.stack()
            dec.decode(tx.lookup(tx.quantize(tx.encode(x[:, :, f * HOP:(f + 1) * HOP]))))
            if t0 is not None:
                n += 1
                if time.perf_counter() - t0 > budget_s:
                    break
    dt = time.perf_counter() - t0
    return {"value": round(n / dt, 2), "unit": "frames/s", "cores": threads, "kind": "port",
            "sample": f"{n} consecutive single-frame (300-sample) encode+RVQ+lookup+v1-decode steps of ONE stream "
                      f"(the reference is batch-1), {dt:.1f} s, torch.set_num_threads({threads}) of {os.cpu_count()} host cpus",
            "ms_per_frame": round(1e3 * dt / n, 2)}


def cpu_cfg1(threads=4, reps=3):
    """BASELINE config 1 on the host cores: demoFile.py:58-61 (one 24000-sample libritts_sym file, one-shot
    encode -> quantize -> lookup -> decode) through the CPU port."""
    from audiodec_amd import synth, configs
    from oracle import audiodec_oracle as O
    torch.set_num_threads(threads)
    _, enc_tag, _, dec_tag, _ = configs.alias("libritts_sym")
    _, _, pe = configs.experiment(enc_tag)
    mt_d, _, pd = configs.experiment(dec_tag)
    tx = O.AutoEncoderOracle(synth.synth_state_dict(enc_tag, SEED), pe, 1)
    zq0 = tx.initial_encoder(8192)
    dec = O.build_decoder_oracle(synth.synth_state_dict(dec_tag, SEED), mt_d, pd, 1)
    dec.initial_decoder(zq0)
    x = torch.from_numpy(synth.synth_audio(SEED, 0, 24000))[None, None, :]
    best = 1e9
    with torch.no_grad():
        for _ in range(reps):
            t0 = time.perf_counter()
            dec.decode(tx.lookup(tx.quantize(tx.encode(x))))
            best = min(best, time.perf_counter() - t0)
    return {"ms": round(1e3 * best, 1), "frames_per_s": round(80 / best, 1), "cores": threads,
            "what": "libritts_sym 24000-sample file, one-shot round trip, CPU port, best of %d" % reps}


def _widen(o, batch):
    """One warmed-up oracle stream -> `batch` identical ones (every stream saw the same silence)."""
    o.batch = batch
    for k in list(o.pad):
        o.pad[k] = o.pad[k].expand(batch, -1, -1).clone()
    return o


def self_check(root, dev, B, fps, serial, steps=2, model=None, decode=True):
    """Parity of a TIMED configuration: a fresh AudioDec with the same stream count, arithmetic, lowering and schedule
    object runs `steps` batches; latent, RVQ indices and waveform are compared with the CPU oracle (the checker, never the
    thing measured).  North-star tolerances: waveform <= 1e-4 max-abs, indices bit-exact.  model / decode: the headline's
    (vctk_v1, whole path) by default; extra_configs passes BASELINE configs 2 (vctk_sym encoder + RVQ only) and 3 (vctk_sym)."""
    from audiodec_amd import synth, configs
    from oracle import audiodec_oracle as O
    model = model or MODEL
    ad = build_audiodec(root, dev, B, fps, model=model)
    pipe = None if serial else TxRxPipeline(ad, dev)
    xs = [torch.from_numpy(np.stack([synth.synth_audio(SEED + 1000 + j, s, HOP * fps) for s in range(B)]))[:, None, :].to(dev)
          for j in range(steps)]
    zs, idxs, ys = [], [], []
    if pipe:
        pipe.enter()
    for x in xs:
        if pipe:
            ys.append(pipe.step(x)); zs.append(pipe.last_z); idxs.append(pipe.last_idx)
        else:
            z = ad.tx_encoder.encode(x); idx = ad.tx_encoder.quantize(z)
            ys.append(ad.decoder.decode(ad.rx_encoder.lookup(idx)) if decode else None); zs.append(z); idxs.append(idx)
    if pipe:
        pipe.exit()
    torch.cuda.synchronize()
    _, enc_tag, _, dec_tag, _ = configs.alias(model)
    mt_d, _, pd = configs.experiment(dec_tag)
    _, _, pe = configs.experiment(enc_tag)
    t0 = time.perf_counter()
    tx = O.AutoEncoderOracle(synth.synth_state_dict(enc_tag, SEED), pe, 1)
    zq0 = tx.initial_encoder(8192)
    dec = O.build_decoder_oracle(synth.synth_state_dict(dec_tag, SEED), mt_d, pd, 1)
    dec.initial_decoder(zq0)
    tx, dec = _widen(tx, B), _widen(dec, B)
    dz = dy = 0.0
    clean = torch.ones(B, dtype=torch.bool)
    flips, unexplained, decisions, min_margin, flip_margins = 0, 0, 0, float("inf"), []
    with torch.no_grad():
        for j in range(steps):
            oz = tx.encode(xs[j].cpu())
            oi, om = tx.quantize(oz, return_margin=True)
            oy = dec.decode(tx.lookup(oi)) if decode else None
            gi = idxs[j].cpu().reshape(oi.shape)
            bad = (gi != oi)
            clean &= ~bad.any(0).any(-1)                          # a stream whose codes differ decodes a different signal from
            dz = max(dz, float((zs[j].cpu() - oz).abs().max()))   # then on: its waveform is compared up to the flip only
            if decode and bool(clean.any()):
                dy = max(dy, float((ys[j].cpu() - oy)[clean].abs().max()))
            for b_, t_ in bad.any(0).nonzero().tolist():          # the first flipped stage of a frame is the decision that differed
                q_ = int(bad[:, b_, t_].nonzero()[0])
                m_ = float(om[q_, b_, t_])
                flips += 1
                flip_margins.append(m_)
                # a flip is explained when the reference's own top-2 distance margin is within reach of the f32 round-off of z
                # (|dz| ~ 1e-6, |E| ~ O(10) -> distance perturbation ~ 1e-5; DESIGN.md, RVQ soak)
                unexplained += m_ >= 1e-4
            decisions += int(oi.numel())
            min_margin = min(min_margin, float(om.min()))
    ok = dz < 1e-4 and dy < 1e-4 and unexplained == 0
    return {"ok": ok, "model": model, "path": "encode -> RVQ -> lookup -> decode" if decode else "encode -> RVQ", "streams": B, "steps": steps,
            "max_abs_dz": dz, "max_abs_dy": dy if decode else None, "indices_equal": flips == 0,
            "frames_with_flipped_indices": flips, "flip_margins": flip_margins,
            "unexplained_flips": int(unexplained), "streams_compared_to_the_end": int(clean.sum()), "rvq_decisions": decisions, "min_reference_top2_margin": min_margin,
            "tolerance": {"waveform_max_abs": 1e-4, "indices": "bit-exact"}, "oracle_cpu_s": round(time.perf_counter() - t0, 1),
            "what": "fresh model, same stream count / arithmetic / program lowering / HIP-stream schedule as the timed run, "
                    "vs the CPU oracle (oracle/audiodec_oracle.py)"}


def self_check_vocoder(root, dev, B, steps=2):
    """Parity of BASELINE config 4 (the v1 vocoder alone, codes -> waveform): a fresh model decodes `steps` batches of seeded random codes;
    lookup sum and waveform against the CPU oracle (lookup: bit-exact -- the same additions in the same order, layers/vq_module.py:159-161;
    waveform <= 1e-4 max-abs)."""
    from audiodec_amd import synth, configs
    from oracle import audiodec_oracle as O
    keep = os.environ.get("ADK_VOCODER_STAGES")
    os.environ["ADK_VOCODER_STAGES"] = "1"
    try:
        ad = build_audiodec(root, dev, B, 1, model="vctk_v1")
    finally:
        if keep is None:
            os.environ.pop("ADK_VOCODER_STAGES", None)
        else:
            os.environ["ADK_VOCODER_STAGES"] = keep
    g = torch.Generator().manual_seed(SEED + 4)
    idxs = [torch.randint(0, 1024, (8, B, 1), generator=g) + 1024 * torch.arange(8).view(8, 1, 1) for _ in range(steps)]
    zqs, ys = [], []
    for idx in idxs:
        zq = ad.rx_encoder.lookup(idx.to(dev))
        zqs.append(zq); ys.append(ad.decoder.decode(zq))
    torch.cuda.synchronize()
    _, enc_tag, _, dec_tag, _ = configs.alias("vctk_v1")
    mt_d, _, pd = configs.experiment(dec_tag)
    _, _, pe = configs.experiment(enc_tag)
    t0 = time.perf_counter()
    tx = O.AutoEncoderOracle(synth.synth_state_dict(enc_tag, SEED), pe, 1)
    zq0 = tx.initial_encoder(8192)
    dec = O.build_decoder_oracle(synth.synth_state_dict(dec_tag, SEED), mt_d, pd, 1)
    dec.initial_decoder(zq0)
    dec = _widen(dec, B)
    dq = dy = 0.0
    with torch.no_grad():
        for j in range(steps):
            ozq = tx.lookup(idxs[j])
            dq = max(dq, float((zqs[j].cpu() - ozq).abs().max()))
            dy = max(dy, float((ys[j].cpu() - dec.decode(ozq)).abs().max()))
    return {"ok": dq == 0.0 and dy < 1e-4, "model": "vctk_v1", "path": "codes -> lookup -> decode", "streams": B, "steps": steps,
            "max_abs_dzq": dq, "max_abs_dy": dy, "oracle_cpu_s": round(time.perf_counter() - t0, 1)}


def extra_configs(root, dev, steps=100, warmup=10, check=True):
    """SURVEY.md 8(d) configs 1-4 on this GPU in the arithmetic of the run (the headline is config 5's per-GPU share).  Configs 2 and 3
    -- whose stream counts (32, 64) select other kernels than the headline's 256: few-streams time tiles, no chain launches -- are also
    CHECKED against the CPU oracle at exactly their stream count (self_check on a fresh model, 3 steps)."""
    def compact(c):
        return {k: c[k] for k in ("ok", "model", "path", "streams", "steps", "max_abs_dz", "max_abs_dy", "indices_equal", "frames_with_flipped_indices",
                                  "unexplained_flips", "min_reference_top2_margin")}

    from audiodec_amd import synth
    from audiodec_amd.audiodec import AudioDec, assign_model
    import contextlib, io

    def load(model, streams, max_frames):
        synth.write_model(root, model, SEED)
        cwd = os.getcwd()
        os.chdir(root)
        try:
            sr, enc, dec = assign_model(model)
            ad = AudioDec(tx_device=dev, rx_device=dev, num_streams=streams, max_frames=max_frames, guard=False)
            with contextlib.redirect_stdout(io.StringIO()):
                ad.load_transmitter(enc)
                ad.load_receiver(enc, dec)
        finally:
            os.chdir(cwd)
        return ad

    def timed(fn, n, w):
        """Seconds per call: warm-up of >= w calls AND >= 0.3 s (a freshly built model follows seconds of host-only work; the first
        ~100 ms of GPU activity after that sporadically contain one stall of 30-80 ms -- tools/hiccup.py: never in steady state),
        then five groups of n / 5 back-to-back calls, each bracketed by a synchronise: the median group."""
        t_w = time.perf_counter()
        i = 0
        while i < w or time.perf_counter() - t_w < 0.3:
            fn()
            i += 1
            if i % 16 == 0:
                torch.cuda.synchronize()
        per = max(1, n // 5)
        groups = []
        for _ in range(5):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(per):
                fn()
            torch.cuda.synchronize()
            groups.append((time.perf_counter() - t0) / per)
        return float(np.median(groups))

    def timed_median(fn, n, w):                    # one-shot latency: each call host-synchronised, median (a box hiccup of tens of ms in
        for _ in range(w):                         # one of 20 calls would otherwise be the whole number)
            fn()
        ts = []
        for _ in range(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            fn()
            torch.cuda.synchronize()
            ts.append(time.perf_counter() - t0)
        return float(np.median(ts)), float(np.max(ts))

    def audio(streams, length):
        return torch.from_numpy(np.stack([synth.synth_audio(SEED, s, length) for s in range(streams)]))[:, None, :].to(dev)

    res = {}
    old_st = os.environ.get("ADK_VOCODER_STAGES")
    os.environ["ADK_VOCODER_STAGES"] = "1"
    try:
        ad = load("libritts_sym", 1, 80)
        x = audio(1, 24000)
        t, t_max = timed_median(lambda: ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x)))), 20, 3)
        res["cfg1_libritts_sym_file_24000_B1"] = {"ms": round(1e3 * t, 3), "ms_max": round(1e3 * t_max, 3), "frames_per_s": round(80 / t, 1),
                                                  "rtf": round(t / 1.0, 5), "what": "median of 20 host-synchronised one-shot round trips"}
        ad = load("vctk_sym", 32, 1)
        x = audio(32, HOP)
        t = timed(lambda: ad.tx_encoder.quantize(ad.tx_encoder.encode(x)), steps, warmup)
        res["cfg2_vctk_encoder_rvq_B32"] = {"ms_per_step": round(1e3 * t, 4), "frames_per_s": round(32 / t, 1)}
        if check:
            res["cfg2_vctk_encoder_rvq_B32"]["self_check"] = compact(self_check(root, dev, 32, 1, True, steps=3, model="vctk_sym", decode=False))
        ad = load("vctk_sym", 64, 1)
        x = audio(64, HOP)
        t = timed(lambda: ad.decoder.decode(ad.rx_encoder.lookup(ad.tx_encoder.quantize(ad.tx_encoder.encode(x)))), steps, warmup)
        res["cfg3_vctk_sym_full_B64"] = {"ms_per_step": round(1e3 * t, 4), "frames_per_s": round(64 / t, 1)}
        if check:
            res["cfg3_vctk_sym_full_B64"]["self_check"] = compact(self_check(root, dev, 64, 1, True, steps=3, model="vctk_sym", decode=True))
        ad = load("vctk_v1", 256, 1)
        g = torch.Generator().manual_seed(SEED)
        idx = (torch.randint(0, 1024, (8, 256, 1), generator=g) + 1024 * torch.arange(8).view(8, 1, 1)).to(dev)
        zq = ad.rx_encoder.lookup(idx)
        t = timed(lambda: ad.decoder.decode(zq), steps, warmup)
        res["cfg4_v1_vocoder_B256"] = {"ms_per_step": round(1e3 * t, 4), "frames_per_s": round(256 / t, 1),
                                       "tflops": round(596.8e6 * 256 / t / 1e12, 2)}
        if check:
            res["cfg4_v1_vocoder_B256"]["self_check"] = self_check_vocoder(root, dev, 256)
        del ad
    finally:
        if old_st is None:
            os.environ.pop("ADK_VOCODER_STAGES", None)
        else:
            os.environ["ADK_VOCODER_STAGES"] = old_st
    res["note"] = "one HIP stream, one program per model half; host-synchronised wall time, median of five groups of steps after >= 0.3 s of warm-up"
    return res


def finish_line(out):
    """Order of the JSON line.  The driver's record keeps the standard keys and the END of stdout, so the figures a reader needs besides
    `value` are repeated, compact, as the LAST key (`summary`): the latency half of the metric, the guard legs, the roofline fractions of
    the dominant kernel and of the north-star's named kernel.  `latency_ms` itself moves to the end too."""
    def g(d, *path):
        for k in path:
            if not isinstance(d, dict) or k not in d:
                return None
            d = d[k]
        return d
    lat = out.pop("latency_ms", None)
    cpu = out.pop("cpu_baseline", None)
    if cpu is not None:
        out["cpu_baseline"] = cpu                  # (verbose: in front of the tail)
    if lat is not None:
        out["latency_ms"] = lat
    ct, t5, roof = out.get("roofline_convtr") or {}, out.get("roofline_convtr_T5") or {}, out.get("roofline") or {}
    out["summary"] = {
        "value_frames_per_s": out.get("value"), "ms_per_step": out.get("ms_per_step"), "n_gpus": out.get("n_gpus"),
        "unguarded": g(out, "unguarded", "value"), "guard_direct_calls": g(out, "guard_direct_calls", "value"),
        "guard_every_step_synchronised": g(out, "guard_synchronous", "value"), "exact_f32": g(out, "other_precision", "value"),
        "latency_ms": {"single_stream": g(lat, "encode_decode_single_stream_median"), "single_stream_guarded_direct_calls": g(out, "guard_direct_calls", "single_stream_ms"),
                       "single_stream_guard_synchronised": g(out, "guard_synchronous", "single_stream_ms"),
                       "single_stream_unguarded_same_loop": g(out, "guard_direct_calls", "single_stream_ms_unguarded_same_loop"),
                       "batch": g(lat, "encode_decode_at_batch_median"), "batch_guarded_direct_calls": g(lat, "encode_decode_at_batch_median_guarded"),
                       "cfg2_encoder_rvq_32_streams": g(out, "extra_configs", "cfg2_vctk_encoder_rvq_B32", "ms_per_step"),
                       "cfg3_sym_codec_64_streams": g(out, "extra_configs", "cfg3_vctk_sym_full_B64", "ms_per_step"),
                       "cfg4_v1_vocoder_256_streams": g(out, "extra_configs", "cfg4_v1_vocoder_B256", "ms_per_step")},
        "roofline": {"kernel": roof.get("kernel"), "frac": roof.get("frac"), "frac_serial": roof.get("frac_serial"), "frac_source": roof.get("frac_source"),
                     "traffic_bytes_per_launch": roof.get("traffic"), "algorithmic_bytes_per_launch": roof.get("algorithmic_bytes_per_launch")},
        "north_star_kernel": {"kernel": ct.get("kernel"), "frac_of_8TBs": ct.get("frac"), "frac_serial": ct.get("frac_serial"), "avg_launch_us_serial": ct.get("avg_launch_us_serial"), "frac_events_serial": ct.get("frac_events_serial"),
                              "T5_frac_serial": t5.get("frac_serial"), "T5_frac": t5.get("frac")},
        "launches_per_step": (roof.get("launches_per_step_all_kernels") + 2) if roof.get("launches_per_step_all_kernels") else None,      # program launches + RVQ search + lookup (no guard-post kernels since ABI 14)
        "north_star_kernel_launch_us": {"serial": ct.get("avg_launch_us_serial"), "pipelined": ct.get("avg_launch_us_pipelined")},
        "self_check_ok": g(out, "self_check", "ok"), "device_error_flags": out.get("device_error_flags"),
    }
    return out


def self_launch(n):
    """Re-exec this command line under `python -m torch.distributed.run --nnodes=1 --nproc-per-node n` (rendezvous on
    127.0.0.1, a free port) and return its exit code.  Fewer than n visible HIP devices is an error (rc 2) unless the
    one-GPU rehearsal hook ADK_BENCH_ONE_GPU=1 is set (tests: every rank on cuda:0 over gloo)."""
    import socket
    import subprocess
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < n and os.environ.get("ADK_BENCH_ONE_GPU") != "1":
        print(f"bench.py --gpus {n}: only {have} HIP device(s) visible", file=sys.stderr)
        return 2
    import __graft_entry__
    __graft_entry__.build()                       # once, before the ranks start (they would serialise on the build lock)
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    return subprocess.call(cmd, env=env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=100)
    ap.add_argument("--warmup", type=int, default=10)
    ap.add_argument("--preroll", type=int, default=64,
                    help="untimed steps BEFORE the warm-up steps: the device sat idle while the models were built, and the first "
                         "tens of milliseconds after idle run at lower clocks (profiles/r3_short_runs.md); reported as "
                         "preroll_steps.  0 = none")
    ap.add_argument("--streams", type=int, default=256, help="streams per GPU")
    ap.add_argument("--frames-per-step", type=int, default=1, help="hops per stream per step (headline: 1)")
    ap.add_argument("--serial", action="store_true", help="one HIP stream (no transmitter/receiver overlap)")
    ap.add_argument("--groups", type=int, default=1, help="split the streams of a GPU into this many independently stepped groups")
    ap.add_argument("--precision", choices=("f32", "split16"), default="split16",
                    help="split16 (default): every matrix-core conv carries each f32 operand as f16 hi + f16 lo/2048 and forms a "
                         "product sum from 3 f16 MFMAs with f32 accumulation (measured error below the f32 MFMA chain, "
                         "profiles/r1_f16_split_probe.txt).  f32: the exact-f32 MFMA kernels everywhere.  The other one is "
                         "timed too and reported under 'other_precision' (single-GPU runs)")
    ap.add_argument("--no-other-precision", action="store_true")
    ap.add_argument("--stages", type=str, default="2",
                    help="vocoder lowering: 1 = one program; 2 = two programs cut in front of upsample stage 2, the second on a "
                         "third HIP stream (default); or explicit cut points, e.g. 1,2 = three programs on three streams")
    ap.add_argument("--graph", choices=("0", "1"), default=os.environ.get("ADK_BENCH_GRAPH", "0"),
                    help="1: replay each program's steady state as HIP graphs (adk_program_set_graph); results are bit-identical")
    ap.add_argument("--pmc-markers", action="store_true",
                    help="launch a recognisable ATen kernel (arange of PMC_MARKER_N elements) just before and just after the timed "
                         "steps, so that tools/pmc_summary.py can tell the steady-state launches of a rocprofv3 --pmc pass from "
                         "model loading, warm-up and the single-stream legs")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-cfg1", action="store_true", help="skip BASELINE config 1 (file round trip) on the host CPU")
    ap.add_argument("--no-self-check", action="store_true", help="skip the parity check of the timed configuration against the CPU oracle")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip SURVEY 8(d) configs 1-4")
    ap.add_argument("--no-op-profile", action="store_true")
    ap.add_argument("--no-t5", action="store_true", help="skip the secondary roofline of the named kernel at the reference streamer's default chunk (5 frames per call)")
    ap.add_argument("--no-guarded", action="store_true", help="skip the legs that time the other guard modes (guard off; every program step checked synchronously)")
    ap.add_argument("--guard", choices=("default", "off"), default="default",
                    help="default: AudioDec's default guard (on; deferred in the pipeline of the timed region).  off: AudioDec(guard=False), as rounds 1-4 timed")
    ap.add_argument("--dump-ops", type=str, default=None, help="write the per-op HIP-event table (CSV) here")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # `python bench.py --gpus N` launches its own N ranks (one process per GPU over RCCL) and relays their exit code;
        # under torchrun (WORLD_SIZE set, the driver's form for N > 1) this branch is not taken
        sys.exit(self_launch(args.gpus))

    import __graft_entry__
    __graft_entry__.build()
    os.environ["ADK_SPLIT16"] = "1" if args.precision == "split16" else "0"    # read by the generators at construction
    os.environ["ADK_VOCODER_STAGES"] = str(args.stages)
    os.environ["ADK_GRAPH"] = args.graph

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus != world:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}")
    assert torch.cuda.is_available(), "bench.py needs a HIP device"
    # test hook (not used by the driver): ADK_BENCH_BACKEND=gloo ADK_BENCH_ONE_GPU=1 runs all ranks on cuda:0 so the
    # multi-rank control flow can be exercised on a 1-GPU box; production is one rank per GPU over RCCL ("nccl")
    backend = os.environ.get("ADK_BENCH_BACKEND", "nccl")
    if os.environ.get("ADK_BENCH_ONE_GPU") == "1":
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = f"cuda:{local_rank}"
    # one rank per GPU, each on its own block of host cpus (ADK_BENCH_PIN=0: leave the affinity alone)
    from audiodec_amd import shard as _shard
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", str(world)))
    pinned = _shard.pin_rank(int(os.environ.get("LOCAL_RANK", "0")), local_world, one_gpu=os.environ.get("ADK_BENCH_ONE_GPU") == "1") if os.environ.get("ADK_BENCH_PIN", "1") != "0" else None
    coll_dev = dev if backend == "nccl" else "cpu"
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device(dev))
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from audiodec_amd import synth, shard, configs
    B = args.streams
    # rank 0 synthesises the checkpoints; the bits reach the other ranks through RCCL (shard.py)
    _, enc_tag, _, dec_tag, _ = configs.alias(MODEL)
    sds = {}
    for tag in (enc_tag, dec_tag):
        sd = synth.synth_state_dict(tag, SEED if rank == 0 else SEED + 1)
        sds[tag] = shard.broadcast_state_dict(sd, src=0, device=coll_dev)
    tmp = tempfile.TemporaryDirectory()
    sr, _, tx_steps, _, rx_steps = configs.alias(MODEL)
    synth.write_experiment(tmp.name, enc_tag, tx_steps, SEED, sd=sds[enc_tag])
    synth.write_experiment(tmp.name, dec_tag, rx_steps, SEED, sd=sds[dec_tag])
    NG = args.groups
    assert B % NG == 0
    FPS = args.frames_per_step
    global BENCH_CONFIG
    BENCH_CONFIG = bench_config_string(B, args.stages, args.frames_per_step, args.precision, args.guard, args.serial)
    guard_arg = {"default": None, "off": False}[args.guard]
    ads = [build_audiodec(tmp.name, dev, B // NG, FPS, guard=guard_arg) for _ in range(NG)]
    ad = ads[0]

    lo, hi = rank * B, (rank + 1) * B            # global stream ids of this rank
    n_buf = 8
    xs = [torch.from_numpy(np.stack([synth.synth_audio(SEED + j, s, HOP * FPS) for s in range(lo, hi)]))[:, None, :].to(dev)
          for j in range(n_buf)]

    def sync_all():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
            torch.cuda.synchronize()

    if NG > 1:
        pipes = [TxRxPipeline(a_, dev) for a_ in ads]
        Bg = B // NG

        class _Multi:
            def enter(self):
                for p_ in pipes: p_.enter()
            def exit(self):
                for p_ in pipes: p_.exit()
            def step(self, x):
                return [p_.step(x[g * Bg:(g + 1) * Bg]) for g, p_ in enumerate(pipes)]
        pipe = _Multi()
    else:
        pipe = None if args.serial else TxRxPipeline(ad, dev)
    run = (lambda x: step(ad, x)) if pipe is None else pipe.step
    with torch.no_grad():
        if args.preroll > 0:                       # wake the device up; its own region, drained before the warm-up starts
            if pipe:
                pipe.enter()
            for i in range(args.preroll):
                run(xs[i % n_buf])
            if pipe:
                pipe.exit()
            sync_all()
        if pipe:
            pipe.enter()
        for i in range(args.warmup):
            run(xs[i % n_buf])
        if pipe:
            pipe.exit()
        sync_all()
        if args.pmc_markers:                       # outside the timed region: see tools/pmc_summary.py
            torch.arange(PMC_MARKER_N, device=dev); torch.cuda.synchronize()
        if hasattr(pipe, "reset_host_times"):
            pipe.reset_host_times()
        t0 = time.perf_counter()
        if pipe:
            pipe.enter()
        for i in range(args.steps):
            y = run(xs[i % n_buf])
        if pipe:
            pipe.exit()
        sync_all()
        elapsed = time.perf_counter() - t0
        if args.pmc_markers:
            torch.arange(PMC_MARKER_N, device=dev); torch.cuda.synchronize()
    elapsed_rank = elapsed
    elapsed = shard.max_over_ranks(elapsed, coll_dev)
    per_rank = shard.gather_floats(B * args.steps * FPS / elapsed_rank, coll_dev)      # every rank's own frames/s over ITS timed region
    # what the Python of a rank costs per step: handing the launches of a batch to the runtime (issue) and reading the deferred checks of
    # older batches (guard; includes waiting for the GPU when the host is `guard_depth` batches ahead) -- every rank's, gathered
    host_issue = shard.gather_floats(1e3 * getattr(pipe, "t_issue", 0.0) / max(1, getattr(pipe, "n_steps", 1)), coll_dev) if hasattr(pipe, "t_issue") else None
    host_guard = shard.gather_floats(1e3 * getattr(pipe, "t_guard", 0.0) / max(1, getattr(pipe, "n_steps", 1)), coll_dev) if hasattr(pipe, "t_issue") else None
    frames = world * B * args.steps * FPS
    ms_per_step = 1e3 * elapsed / args.steps
    out = {
        "metric": "48 kHz hop-300 frames/s/GPU + per-frame encode+decode latency (ms)",
        "value": round(frames / elapsed, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps,
        "warmup": args.warmup, "preroll_steps": args.preroll, "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak",
        "vs_baseline": None,
        "latency_ms": {},                      # (the second half of the metric: filled below, kept early in the line)
        "dtype": "f32" if args.precision == "f32" else "f32 (conv operands split into f16 hi + lo/2048 pairs, 3 f16 MFMAs per product sum, f32 accumulate)",
        "data": "synthetic",
        "config": {"workload": f"{MODEL} full pipeline (symAD encoder+projector -> 8x1024 RVQ -> lookup -> AudioDec-v1 "
                               "HiFi-GAN vocoder), 48 kHz hop 300, streaming, 1 frame per stream per step "
                               "(BASELINE.json config 5 per-GPU share)",
                   "streams_per_gpu": B, "streams_total": world * B, "frames_per_step_per_stream": FPS,
                   "sample_rate": 48000, "hop": HOP, "weights": "seeded synthetic (audiodec_amd/synth.py), fp32",
                   "schedule": "serial, one HIP stream" if args.serial else
                               ("transmitter (encode+RVQ) and receiver (lookup+vocoder) on two HIP streams, codes handed over by event" if args.stages == "1" else
                                f"software pipeline over batches on HIP streams, handed over by events: encode+RVQ | lookup + vocoder programs cut in front of upsample stage(s) {'2' if args.stages == '2' else args.stages}")},
        "precision": {"mode": args.precision,
                      "note": "split16: v = hi + lo/2048 with hi = f16(v), lo = f16((v - hi)*2048); sum(a*b) = sum(a_hi*b_hi) + "
                              "(sum(a_hi*b_lo) + sum(a_lo*b_hi))/2048 on v_mfma_f32_32x32x16_f16 with f32 accumulators; measured max "
                              "|err|/sum|a b| vs fp64 8e-8 (f32 MFMA chain: 2e-7), profiles/r1_f16_split_probe.txt; both modes pass "
                              "the same parity tests (waveform <= 1e-4 vs the reference, RVQ indices bit-exact)"
                              if args.precision == "split16" else "exact-f32 MFMA (v_mfma_f32_32x32x2_f32) everywhere"},
        "executor": ("HIP graphs: one captured launch sequence per program and cursor phase, replayed by hipGraphLaunch (ops on caller buffers stay "
                     "ordinary launches)") if args.graph == "1" else "ordinary launches from the C++ program runner (one C call per program and step)",
        "guard": ("off (--guard off: AudioDec(guard=False); device flag words checked once after the run and asserted 0)" if args.guard == "off" else
                  "on, the default of AudioDec(...), for `value`: every program step of every batch is checked on the device and a split-f16 range "
                  "overflow is repaired by replay on the exact-f32 kernels -- deferred: the check of a batch is read (non-blocking) at the entry of a "
                  "later step, at most `guard_depth` batches are unverified (audiodec_amd/pipeline.py); `unguarded` = the same with guard=False, "
                  "`guard_direct_calls` = direct calls of the facade in their default guard mode (checked when a result is first looked at: "
                  "audiodec_amd/lazy_guard.py), `guard_synchronous` = every program step checked before the next is issued (ADK_GUARD_MODE=sync)"),
        "guard_depth": getattr(pipe, "depth", None) if NG == 1 else None,
        "guard_stats": ({"batches_verified": pipe.log.verified, "host_waits_for_the_oldest_batch": pipe.log.waits, "repairs": pipe.log.repairs}
                        if NG == 1 and getattr(pipe, "log", None) is not None else None),
        "frames_per_s_per_gpu": round(frames / elapsed / world, 1),
        "frames_per_s_of_each_rank": [round(v, 1) for v in per_rank],
        "host_ms_per_step_of_each_rank": None if host_issue is None else {
            "issue": [round(v, 4) for v in host_issue], "guard_polls_and_waits": [round(v, 4) for v in host_guard],
            "cpus_of_rank_0": (f"{pinned[0]}-{pinned[-1]} ({len(pinned)} cpus, pinned)" if pinned else f"not pinned ({os.cpu_count()} host cpus)"),
            "note": "issue = Python + runtime time to enqueue one batch (all programs, all HIP streams); must stay well below ms_per_step for the host not to be "
                    "the bottleneck -- with N ranks on one host each rank is pinned to 1/N of the cpus (audiodec_amd/shard.py: pin_rank)"},
        "distributed": {"world_size": world, "backend": (backend + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else None,
                        "ranks_per_gpu": (world if os.environ.get("ADK_BENCH_ONE_GPU") == "1" else 1), "steady_state_collectives": 0,
                        "note": "one process per GPU, streams [r*B, (r+1)*B) on rank r, full weight replica per rank; the collectives of a run are the "
                                "checkpoint broadcast, the barrier around the timed region and the max / gather of the elapsed times"},
        "realtime_streams_supported_per_gpu": int(frames / elapsed / world / 160.0),
    }

    # per-batch latency: device-complete time of ONE encode -> RVQ -> lookup -> decode pass, serial, synchronised
    def batch_latency():
        lat = []
        for i in range(12):
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for g_, a_ in enumerate(ads):
                step(a_, xs[i % n_buf][g_ * (B // NG):(g_ + 1) * (B // NG)])
            torch.cuda.synchronize()
            lat.append(1e3 * (time.perf_counter() - t1))
        return round(float(np.median(lat[2:])), 4)

    with torch.no_grad():
        with unguarded(*ads):
            out["latency_ms"]["encode_decode_at_batch_median"] = batch_latency()
        if args.guard == "default":
            out["latency_ms"]["encode_decode_at_batch_median_guarded"] = batch_latency()      # direct calls with the default guard (lazy: checked when a result is looked at / by a later call)

    if rank == 0:
        # rank 0 prices its own GPU's kernels at every N (a few seconds); the legs below it are single-GPU extras
        with torch.no_grad():
            if not args.no_op_profile and NG == 1:
                with unguarded(ad):
                    rows = op_profile(ad, xs, B, 10, FPS)                              # serial: one HIP stream, nothing else on the chip
                rows_pipe = op_profile(ad, xs, B, 10, FPS, pipe=pipe) if pipe is not None else None      # the schedule `value` was timed in
                if args.dump_ops:
                    with open(args.dump_ops, "w") as f:
                        f.write("prog,op,kernel,cin_g,cout_g,groups,taps,dil,t_out_per_stream,us,gflop,tflops,us_pipelined\n")
                        for qi, r in enumerate(rows):
                            c = r["op"].conv
                            f.write(f"{r['prog']},{r['name']},{r['kernel']},{c.cin_g},{c.cout_g},{c.groups},{c.taps},{c.dilation},"
                                    f"{r['op'].rate_out},{1e3 * r['ms']:.2f},{r['flops'] / 1e9:.3f},"
                                    f"{(r['flops'] / (r['ms'] * 1e-3) / 1e12) if r['ms'] > 0 else 0:.2f},"
                                    f"{(1e3 * rows_pipe[qi]['ms']) if rows_pipe is not None else float('nan'):.2f}\n")
                roof, roof_ct, kernels = roofline_from(rows, B, FPS, args.precision == "split16", rows_pipe)
                if roof_ct is not None:
                    # the transposed conv of the named kernel BY ITSELF (conv_up16, 300 back-to-back launches): the figure of rounds 1-2, kept for
                    # comparison with the fused launch above
                    us, kname = convtr_standalone(dev, sds[dec_tag], B, FPS, args.precision == "split16")
                    b_alone = 4.0 * (64 * (100 * FPS + 1) + 32 * 300 * FPS) * B + 4.0 * 64 * 32 * 6
                    roof_ct["transposed_conv_alone"] = {"kernel": kname, "avg_launch_us": round(us, 2), "bytes_per_launch": b_alone,
                                                        "achieved": round(b_alone / (us * 1e-6) / 1e9, 1), "frac": round(b_alone / (us * 1e-6) / 1e9 / HBM_PEAK_GBS, 4),
                                                        "how": "300 back-to-back launches of upsamples.3 alone (same weights, 256 streams x 100 input steps), HIP events inside the library"}
                out["roofline"] = roof
                out["roofline_convtr"] = roof_ct
                out["kernels"] = kernels
                if rank == 0 and world == 1 and FPS == 1 and not args.no_t5:
                    out["roofline_convtr_T5"] = convtr_t5(tmp.name, dev, sds[dec_tag], B, args.precision == "split16", check=not args.no_self_check)
                enc_ms = sum(r["ms"] for r in rows if r["prog"] == "encoder")
                dec_ms = sum(r["ms"] for r in rows if r["prog"].startswith("decoder"))
                out["latency_ms"]["encoder_kernels_at_batch"] = round(enc_ms, 4)
                out["latency_ms"]["decoder_kernels_at_batch"] = round(dec_ms, 4)
                tot_flops = sum(r["flops"] for r in rows)
                out["pipeline_tflops"] = round(tot_flops / (ms_per_step * 1e-3) / 1e12, 2)
    if rank == 0 and world == 1:
        with torch.no_grad():
            # single-stream latency: device-complete time of one encode+decode step, B = 1
            # (the facade's default lowering -- the vocoder as ONE program: the cut in front of stage 2 exists for the three-stream
            # pipeline above and costs a lone stream one more launch, the copy of the stage boundary, and under the guard one more check)
            def build_single(guard):
                keep = os.environ.get("ADK_VOCODER_STAGES")
                os.environ["ADK_VOCODER_STAGES"] = "1"
                try:
                    return build_audiodec(tmp.name, dev, 1, 1, guard=guard)
                finally:
                    if keep is None:
                        os.environ.pop("ADK_VOCODER_STAGES", None)
                    else:
                        os.environ["ADK_VOCODER_STAGES"] = keep
            ad1 = build_single(False)
            x1 = xs[0][:1, :, :HOP].contiguous()
            for _ in range(10):
                step(ad1, x1)
            torch.cuda.synchronize()
            lat = []
            for _ in range(50):
                t1 = time.perf_counter()
                step(ad1, x1)
                torch.cuda.synchronize()
                lat.append(1e3 * (time.perf_counter() - t1))
            out["latency_ms"]["encode_decode_single_stream_median"] = round(float(np.median(lat)), 4)
            out["latency_ms"]["encode_decode_single_stream_min"] = round(float(np.min(lat)), 4)
            out["latency_ms"]["note"] = ("one 300-sample frame per stream per call; x on device -> y on device, host-synchronised; at_batch: the timed model (vocoder "
                                         "lowered as the pipeline's programs); single_stream: a one-stream model in the facade's default lowering (vocoder = one program)")
            if not args.no_guarded and NG == 1:
                # The same workload in the other guard modes, same streams / schedule object / inputs.  `unguarded`: AudioDec(guard=False), nothing
                # checked per step -- what rounds 1-4 timed as `value`; the difference to `value` is the price of the deferred guard (one 1-thread
                # kernel + an event per program step, the polls, rings with guard_depth steps of extra rows).  `guard_synchronous`: every program
                # step is followed by adk_program_flags (one 1-thread kernel + a stream synchronisation) before the next is issued -- what a direct
                # caller of encode() / decode() gets, and what the pipeline did with the guard on until round 4: its HIP streams can no longer
                # overlap batches.
                def timed_pipeline(guard, depth):
                    adg = build_audiodec(tmp.name, dev, B, FPS, guard=guard)
                    pg = None if args.serial else TxRxPipeline(adg, dev, depth=depth)
                    rung = (lambda x: step(adg, x)) if pg is None else pg.step
                    ng = max(20, args.steps // 2)
                    if pg:
                        pg.enter()
                    for i in range(5 + args.preroll // 2):
                        rung(xs[i % n_buf])
                    if pg:
                        pg.exit()
                    torch.cuda.synchronize()
                    regions = []
                    for _ in range(3):                     # three timed regions, the median: a freshly built model sporadically sees one stall of tens of
                        tg = time.perf_counter()           # milliseconds in its first 100 ms of GPU work (profiles/r2_step_time_outliers.log), which is one of
                        if pg:                             # these short regions whole
                            pg.enter()
                        for i in range(ng):
                            rung(xs[i % n_buf])
                        if pg:
                            pg.exit()
                        torch.cuda.synchronize()
                        regions.append(time.perf_counter() - tg)
                    eg = float(np.median(regions))
                    return {"value": round(B * ng * FPS / eg, 1), "unit": "frames/s", "ms_per_step": round(1e3 * eg / ng, 4), "steps": ng,
                            "regions_ms_per_step": [round(1e3 * e_ / ng, 4) for e_ in regions]}
                out["unguarded"] = timed_pipeline(False, 4)
                out["unguarded"]["what"] = "AudioDec(guard=False): the same workload and schedule with no per-step check (rounds 1-4 timed this as `value`)"

                # single stream in the three guard modes, INTERLEAVED (five rounds of 20 host-synchronised steps per model, median over all): models
                # measured one after the other differed by +-25 us from run to run (clock / allocator state), more than the guard costs
                def build_mode(mode):
                    keep_mode = os.environ.get("ADK_GUARD_MODE")
                    if mode == "sync":
                        os.environ["ADK_GUARD_MODE"] = "sync"          # read by the generators at construction
                    try:
                        return build_single(False if mode == "off" else True)
                    finally:
                        if keep_mode is None:
                            os.environ.pop("ADK_GUARD_MODE", None)
                        else:
                            os.environ["ADK_GUARD_MODE"] = keep_mode
                singles = {m_: build_mode(m_) for m_ in ("off", "lazy", "sync")}
                lat_m = {m_: [] for m_ in singles}
                for a_ in singles.values():
                    for _ in range(10):
                        step(a_, x1)
                torch.cuda.synchronize()
                for _ in range(5):
                    for m_, a_ in singles.items():
                        for _ in range(20):
                            t1 = time.perf_counter()
                            step(a_, x1)
                            torch.cuda.synchronize()
                            lat_m[m_].append(1e3 * (time.perf_counter() - t1))
                ss = {m_: (round(float(np.median(v_)), 4), round(float(np.min(v_)), 4)) for m_, v_ in lat_m.items()}
                lg_ = singles["lazy"].tx_encoder._log
                # direct calls of the drop-in surface (encode / quantize / lookup / decode, one after the other) in the DEFAULT guard mode of
                # round 6 ("lazy", audiodec_amd/lazy_guard.py): the check of a call is posted behind it and read when its result is first looked
                # at, or by a later call; here nothing looks (as in the timed region of `value`): the calls of a batch go out on the pipeline object's
                # three HIP streams with depth = 0, i.e. the pipeline's own deferred guard is OFF and the generators' call log does the work
                out["guard_direct_calls"] = timed_pipeline(None, 0)
                out["guard_direct_calls"].update({
                    "single_stream_ms": ss["lazy"][0], "single_stream_ms_min": ss["lazy"][1], "single_stream_ms_unguarded_same_loop": ss["off"][0],
                    "mode": singles["lazy"].tx_encoder.guard_mode, "calls_verified_single_stream": lg_.verified if lg_ is not None else None,
                    "host_waits_single_stream": lg_.waits if lg_ is not None else None,
                    "what": "guard on (AudioDec's default), direct calls: every program step posts an event behind it, results come back as GuardedTensor "
                            "and are verified when first looked at (.cpu(), .to(), data_ptr(), any torch op) or by a later call; nothing waits per step.  "
                            "single_stream_ms*: one-stream models in the three guard modes measured interleaved (5 x 20 host-synchronised steps each)"})
                del singles
                keep_mode = os.environ.get("ADK_GUARD_MODE")
                os.environ["ADK_GUARD_MODE"] = "sync"              # read by the generators at construction
                try:
                    out["guard_synchronous"] = timed_pipeline(True, 0)
                finally:
                    if keep_mode is None:
                        os.environ.pop("ADK_GUARD_MODE", None)
                    else:
                        os.environ["ADK_GUARD_MODE"] = keep_mode
                ss = (ss["sync"][0], ss["sync"][1])
                out["guard_synchronous"].update({
                    "single_stream_ms": ss[0], "single_stream_ms_min": ss[1],
                    "what": "guard on, mode 'sync' (ADK_GUARD_MODE=sync: the direct-call guard of rounds 3-5): every program step checked before the next one is "
                            "issued (adk_program_flags: one stream synchronisation + one host-side exchange of the flag word per program and step)"})
            if not args.no_other_precision and NG == 1:
                # the same workload through the other arithmetic (same weights, same inputs, same schedule)
                other = "f32" if args.precision == "split16" else "split16"
                os.environ["ADK_SPLIT16"] = "1" if other == "split16" else "0"
                ad2 = build_audiodec(tmp.name, dev, B, FPS)
                os.environ["ADK_SPLIT16"] = "1" if args.precision == "split16" else "0"
                pipe2 = None if args.serial else TxRxPipeline(ad2, dev)
                run2 = (lambda x: step(ad2, x)) if pipe2 is None else pipe2.step
                n2 = max(20, args.steps // 2)
                if pipe2:
                    pipe2.enter()
                for i in range(5 + args.preroll):          # (built on the host while the device idled: same pre-roll as the headline)
                    run2(xs[i % n_buf])
                if pipe2:
                    pipe2.exit()
                torch.cuda.synchronize()
                t2 = time.perf_counter()
                if pipe2:
                    pipe2.enter()
                for i in range(n2):
                    y2 = run2(xs[i % n_buf])
                if pipe2:
                    pipe2.exit()
                torch.cuda.synchronize()
                e2 = time.perf_counter() - t2
                out["other_precision"] = {"precision": other, "value": round(B * n2 * FPS / e2, 1), "unit": "frames/s",
                                          "ms_per_step": round(1e3 * e2 / n2, 4), "steps": n2}
                del ad2
            if not args.no_self_check and NG == 1:
                out["self_check"] = self_check(tmp.name, dev, B, FPS, args.serial)
            if not args.no_extra_configs:
                out["extra_configs"] = extra_configs(tmp.name, dev, check=not args.no_self_check)
        if not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline()
            ncpu = os.cpu_count() or 1
            many = min(ncpu, 16)
            # more legs of the same measurement: the reference's per-call inspect.stack() cost emulated on top
            # of the port, and all host cores instead of the demo's 4 threads (demoFile.py:28)
            out["cpu_baseline"]["as_is_emulated"] = {k: v for k, v in cpu_baseline(6.0, 4, stack_calls=6).items() if k in ("value", "unit", "cores", "sample", "ms_per_frame")}
            out["cpu_baseline"]["as_is_emulated"]["note"] = ("the port + 6 inspect.stack() calls per frame = the reference's check_mode call sites on this path "
                                                            "(encoder.py:77,138, projector.py:53); the unmodified reference cannot run on the GPU box (no /root/reference there)")
            out["cpu_baseline"]["more_cores"] = {k: v for k, v in cpu_baseline(6.0, many).items() if k in ("value", "unit", "cores", "sample", "ms_per_frame")}
            out["cpu_baseline"]["more_cores"]["note"] = (f"{many} of the host's {ncpu} cpus: a batch-1 frame is ~70 small convolutions, more intra-op threads do not help "
                                                         "-- with all 256 threads of the GPU box one frame took 24.5 s (oversubscribed ATen thread pool; measured in round 2, "
                                                         "profiles/r2_cpu_legs.md), so the all-cores leg is capped here to keep the default run bounded")
            if not args.no_cpu_cfg1:
                out["cpu_baseline"]["config1_file_roundtrip"] = cpu_cfg1()
                out["cpu_baseline"]["config1_file_roundtrip_more_cores"] = cpu_cfg1(threads=many, reps=2)
            torch.set_num_threads(4)
    # sticky device-side error flags of the HIP library (0 = clean; see adk_debug_flags in the header)
    import ctypes
    from audiodec_amd import native
    flags = ctypes.c_int32(0)
    native.check(native.lib().adk_debug_flags(ctypes.byref(flags)), "adk_debug_flags")
    out["device_error_flags"] = int(flags.value)
    assert flags.value == 0, f"device error flags {flags.value}: results invalid"
    if "self_check" in out:
        assert out["self_check"]["ok"], f"parity check of the timed configuration failed: {out['self_check']}"
    if "self_check" in out.get("roofline_convtr_T5", {}):
        assert out["roofline_convtr_T5"]["self_check"]["ok"], f"parity check of the 5-frames-per-call configuration failed: {out['roofline_convtr_T5']['self_check']}"
    for k_, v_ in out.get("extra_configs", {}).items():
        if isinstance(v_, dict) and "self_check" in v_:
            assert v_["self_check"]["ok"], f"parity check of {k_} failed: {v_['self_check']}"
    if rank == 0:
        print(json.dumps(finish_line(out)))
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
