timeout 1200 python -m pytest tests -m gpu -x -q 2>&1 | tail -6
for rep in 1 2; do for sh in 0 1; do
ADK_SHADOWS=$sh timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('shadows', $sh, d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encoder_kernels_at_batch'], d['latency_ms']['decoder_kernels_at_batch'], d['latency_ms']['encode_decode_single_stream_median'], d['device_error_flags'])"
done; done
