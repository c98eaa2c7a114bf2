for cfg in "1 256" "2 128" "2 192" "2 256"; do set -- $cfg
ADK_BENCH_WORKGROUPS=$2 timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-other-precision --no-op-profile --groups $1 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('groups', $1, 'wg', $2, d['value'], d['ms_per_step'])"; done
