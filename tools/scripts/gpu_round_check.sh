set -x
mkdir -p gpurun_out
python -c "import __graft_entry__ as g; g.build(); g.smoke()" 2>&1 | tail -3
timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
timeout 400 python bench.py > gpurun_out/bench_default.json 2> gpurun_out/bench_default.err; tail -2 gpurun_out/bench_default.err; python - <<'PY'
import json
d=json.load(open('gpurun_out/bench_default.json'))
for k in ('value','ms_per_step','dtype','latency_ms','roofline','roofline_convtr','other_precision','cpu_baseline','pipeline_tflops','device_error_flags'): print(k, d.get(k))
for k,v in d.get('kernels',{}).items(): print('  ',k,v)
PY
ADK_SPLIT16=1 timeout 300 python tools/config_bench.py --out gpurun_out/configs_split16.json > gpurun_out/configs_split16.log 2>&1; tail -22 gpurun_out/configs_split16.log
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/prof_r1f -o bench -- python $GRAFT_REPO_ROOT/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-other-precision > $GRAFT_REPO_ROOT/gpurun_out/prof_r1f_bench.log 2>&1
cd $GRAFT_REPO_ROOT
ls gpurun_out/prof_r1f | head
