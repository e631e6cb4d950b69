#!/bin/bash
# Instruction-cache counters for the bench kernels (one PMC pass of its own).  Usage (GPU box): bash tools/pmc_icache.sh <outdir> [bench args]
set -e
OUT=${1:-gpurun_out/pmc_icache}; shift || true
ARGS=${@:-"--steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-host-legs"}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p "$OUT"
rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQ_IFETCH SQ_WAVE_CYCLES --output-format csv -d "$OUT/ic" -- python3 bench.py $ARGS > "$OUT/ic.json" 2> "$OUT/ic.err" || echo "pass failed"
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections
out = sys.argv[1]
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for f in glob.glob(out + "/ic/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        k = r["Kernel_Name"].split("(")[0][:60]
        acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); 
        if r["Counter_Name"] == "SQ_WAVE_CYCLES": n[k] += 1
for k, v in acc.items():
    if "blind_rotate" in k or "keyswitch" in k:
        req = v.get("SQC_ICACHE_REQ", 0) or 1
        print(f"{k}: launches {n[k]}  ICACHE_REQ {req:.4g}  HITS {v.get('SQC_ICACHE_HITS',0):.4g}  MISSES {v.get('SQC_ICACHE_MISSES',0):.4g} ({100*v.get('SQC_ICACHE_MISSES',0)/req:.2f} %)  DUP {v.get('SQC_ICACHE_MISSES_DUPLICATE',0):.4g}  IFETCH {v.get('SQ_IFETCH',0):.4g}  WAVE_CYCLES {v.get('SQ_WAVE_CYCLES',0):.4g}")
PY
