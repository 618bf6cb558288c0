for r in 40 48 56 64 72 80 90 98 108 120 134 180; do
  DS_FUSED2_ROWS=$r timeout -k 10 100 python -u bench.py --workload 1080p_128 --steps 101 --warmup 10 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows', $r, round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],3))" || exit 1
done
