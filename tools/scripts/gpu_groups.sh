for g in 1 2 4; do
timeout 300 python bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-op-profile --precision split16 --groups $g 2>/dev/null | python -c "
import sys, json
d=json.loads(sys.stdin.read()); print('groups', $g, d['value'], d['ms_per_step'])"
done
