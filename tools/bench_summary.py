"""One-screen summary of a bench.py line: python tools/bench_summary.py <bench.json>  (value, sampled clock, secondary legs)."""
import json
import sys

d = json.load(open(sys.argv[1]))
s = d.get("secondary", {})
print(d["value"], d["clock"]["sclk_mhz_under_load"], "roofline", d["roofline"]["frac"])
for k, v in s.items():
    if isinstance(v, dict):
        for kk in ("gates_per_s", "bootstraps_per_s", "pairs_per_s", "pairs_per_s_over_adder8", "ripple_as_written",
                   "ripple_rewritten", "prefix_log_depth", "prefix_over_ripple"):
            if kk in v:
                print(" ", k, kk, v[kk])
