# developer tool: the benchmark twice on the same box, without and with an environment setting
# usage: ab_bench.sh <workload> <steps> VAR=value
w=$1; st=$2; kv=$3
for rep in 1 2; do
for mode in base "$kv"; do
  if [ "$mode" = base ]; then pre=""; else pre="$kv"; fi
  env $pre timeout -k 10 100 python -u bench.py --workload $w --steps $st --warmup 6 --no-cpu-baseline --no-companions 2>/dev/null | python3 -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$mode', round(d['value'],1), round(d['ms_per_step'],4), round(d['roofline']['avg_launch_ms'],3))" || exit 1
done; done
