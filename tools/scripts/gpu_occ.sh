for occ in 2 3 4; do for mu in 1 2 4; do
ADK_CONV_OCC=$occ ADK_CONV_MIN_UNITS=$mu timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision --stages 2 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('occ', $occ, 'min_units', $mu, d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encode_decode_single_stream_median'], d['kernels'].get('conv_sk16<64x64>'))"
done; done
