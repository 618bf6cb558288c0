# developer tool: the benchmark with N rows per block of the two-iteration kernel
# usage: sweep_fused2_rows.sh <workload> <steps> N...
w=$1; st=$2; shift 2
for r in "$@"; do
  DS_FUSED2_ROWS=$r timeout -k 10 100 python -u bench.py --workload $w --steps $st --warmup 6 --no-cpu-baseline 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('rows', $r, round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],3))" || exit 1
done
