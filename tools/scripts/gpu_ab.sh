for rep in 1 2; do for lib in prev cur; do
if [ $lib = prev ]; then export ADK_LIB_PATH=$PWD/tools/bin/libadk_prev.so; else unset ADK_LIB_PATH; fi
timeout 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --no-other-precision --no-op-profile 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('$lib', d['value'], d['ms_per_step'], d['latency_ms']['encode_decode_at_batch_median'], d['latency_ms']['encode_decode_single_stream_median'], d['device_error_flags'])"
done; done
