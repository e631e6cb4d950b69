"""One-screen summary of a bench.py line: python tools/bench_summary.py <bench.json>  (value, sampled clock, secondary legs)."""
import json,sys
d=json.load(open(sys.argv[1])); s=d["secondary"]
print(d["value"], d["clock"]["sclk_mhz_under_load"])
for k,v in s.items():
    if isinstance(v,dict):
        for kk in ("gates_per_s","bootstraps_per_s"):
            if kk in v: print(" ",k,kk,v[kk])
