"""Print ms_per_step, value and the roofline line of a bench.py JSON line read from stdin."""
import json
import sys

d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d.get("roofline", {})
print("ms_per_step %.3f  value %.1f  frac %.3f  avg_launch_ms %.3f  placement %s"
      % (d["ms_per_step"], d["value"], r.get("frac", 0), r.get("avg_launch_ms", 0), r.get("placement")))
