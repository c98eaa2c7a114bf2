for kd in 1 2; do echo KD=$kd; ADK_SK16_KD=$kd timeout 300 python tools/conv_bench.py --shape s0,e2,e3,up0,d3,p --impl 6 --cfg=2 --check 2>&1 | grep -v amdgpu.ids; done
ADK_SK16_KD=2 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "split16 or conv_kernels or pipeline" 2>&1 | tail -3
for kd in 1 2; do ADK_SK16_KD=$kd timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('KD', $kd, d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encode_decode_single_stream_median'], d['kernels'].get('conv_sk16<64x64>'), d['device_error_flags'])"; done
