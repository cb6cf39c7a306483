"""Pretty-print the one-line JSON of bench.py (stdin) for tuning sweeps."""
import json, sys
tag = sys.argv[1] if len(sys.argv) > 1 else ""
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
r = d["roofline"]; tu = r["tree_update"]
print(tag, d["config"]["workload"][:20], "it/s %.1f" % d["value"], "frac %.3f" % r["frac"], "k_tree %.1f" % tu["k_tree_us"],
      "k_control %.1f" % tu["k_control_us"], "sweep_us %.0f" % r["sweep_wall_us"])
