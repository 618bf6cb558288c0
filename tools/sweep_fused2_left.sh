# developer tool: the benchmark with the first N columns left to the one-iteration kernel
# usage: sweep_fused2_left.sh <workload> <steps> N...
w=$1; st=$2; shift 2
for c in "$@"; do
  DS_FUSED2_LEFT=$c timeout -k 10 100 python -u bench.py --workload $w --steps $st --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('left', $c, round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],3))" || exit 1
done
