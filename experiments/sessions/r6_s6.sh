#!/bin/bash
# Round 6, session 6: what the lazy guard costs a single stream (host-synchronised steps; guard off / lazy / sync, alternating)
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
for r in 1 2; do for g in off lazy sync; do ADK_SS_GUARD=$g timeout 300 python tools/single_stream_steps.py 1 200 2>&1 | grep "median"; done; done
python - <<'PY'
import os, sys, time, tempfile
sys.path.insert(0, os.getcwd())
os.environ["ADK_VOCODER_STAGES"] = "1"
import numpy as np, torch, bench
from audiodec_amd import synth
dev = torch.device("cuda:0"); root = tempfile.mkdtemp(); synth.write_model(root, bench.MODEL, bench.SEED)
x = torch.from_numpy(np.stack([synth.synth_audio(5, 0, bench.HOP)]))[:, None, :].to(dev)
for g in (False, True):
    ad = bench.build_audiodec(root, dev, 1, 1, guard=g)
    with torch.no_grad():
        for _ in range(20): bench.step(ad, x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(200): bench.step(ad, x)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    lg = ad.tx_encoder._log
    print(f"guard {g}: host enqueue {1e3*(t1-t0)/200:.4f} ms/step, device-complete {1e3*(t2-t0)/200:.4f} ms/step", (lg.verified, lg.waits) if lg else None)
PY
