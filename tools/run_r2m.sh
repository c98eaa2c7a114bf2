#!/bin/bash
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
K=tools/bin/kbench
{
echo "== kbench: register budget 3 waves/SIMD (lb3) / 4 (lb4) vs default, workgroups per CU 2|3|4 =="
for s in s0 s1 e3 up0 d3; do
  $K conv $s 4 256 100
  for occ in 2 3; do ADK_CONV_OCC=$occ LD_LIBRARY_PATH=tools/bin/lb3 $K conv $s 4 256 100; done
  for occ in 2 4; do ADK_CONV_OCC=$occ LD_LIBRARY_PATH=tools/bin/lb4 $K conv $s 4 256 100; done
done
} > gpurun_out/r2m_kbench.log 2>&1
cat gpurun_out/r2m_kbench.log
X="--steps 200 --warmup 20 --no-cpu-baseline --no-extra-configs --no-self-check --no-other-precision --no-op-profile"
python bench.py $X > gpurun_out/r2m_base.json 2>/dev/null
ADK_LIB_PATH=$PWD/tools/bin/lb3/libaudiodec_hip.so python bench.py $X > gpurun_out/r2m_lb3.json 2>/dev/null
ADK_LIB_PATH=$PWD/tools/bin/lb3/libaudiodec_hip.so ADK_BENCH_WORKGROUPS=384 python bench.py $X > gpurun_out/r2m_lb3_wg384.json 2>/dev/null
ADK_LIB_PATH=$PWD/tools/bin/lb3/libaudiodec_hip.so ADK_BENCH_WORKGROUPS=512 python bench.py $X > gpurun_out/r2m_lb3_wg512.json 2>/dev/null
ADK_LIB_PATH=$PWD/tools/bin/lb4/libaudiodec_hip.so python bench.py $X > gpurun_out/r2m_lb4.json 2>/dev/null
ADK_LIB_PATH=$PWD/tools/bin/lb3/libaudiodec_hip.so python bench.py $X --stages 1,2 > gpurun_out/r2m_lb3_st12.json 2>/dev/null
python - <<'PY'
import json,glob
for f in sorted(glob.glob("gpurun_out/r2m_*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"], d["latency_ms"].get("encode_decode_at_batch_median"), d["latency_ms"].get("encode_decode_single_stream_median"))
    except Exception as e: print(f,"ERR",e)
PY
