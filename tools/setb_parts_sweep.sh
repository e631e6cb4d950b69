#!/bin/bash
# VERDICT r5 task 5: Set B's blind rotation is cut into EOC_TFHE_BR_PARTS consecutive launches so that the 96 KB of key rows
# a step reads stay inside an XCD's L2 share while all workgroups walk the same steps.  Sweep parts x slice (both are
# environment knobs: no build) at 1024 and 16 384 gates, then the L2 counters (FETCH_SIZE, TCC_HIT / TCC_MISS: separate
# rocprofv3 --pmc passes) at 1024 gates.  Usage: gpurun -- 'bash tools/setb_parts_sweep.sh > gpurun_out/r06_setb_parts.txt'
cd /tmp && export TMPDIR=/tmp && cd "${GRAFT_REPO_ROOT:-.}"
O=gpurun_out/setb_parts
mkdir -p "$O"
one() { # parts slice gates steps
  EOC_TFHE_BR_PARTS=$1 EOC_TFHE_BR_SLICE=$2 python3 bench.py --pset B --gates $3 --steps $4 --warmup 3 --no-cpu-baseline --no-secondary --no-host-legs 2>/dev/null |
    python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('%.4f ms/step  %.1f gates/s  %d MHz %s' % (d['ms_per_step'], d['value'], d['clock']['sclk_mhz_under_load'], 'ok' if d['decrypt_ok'] else 'WRONG-RESULT'))"
}
echo "== times (Set B, NAND, operands resident; slice 0 = the resident set, 1024 jobs)"
for pass in 1 2; do
  for parts in 1 2 3 4; do
    for slice in 0 768; do
      echo "pass $pass parts=$parts slice=$slice  1024 gates: $(one $parts $slice 1024 30)   16384 gates: $(one $parts $slice 16384 4)"
    done
  done
done
echo "== L2 counters at 1024 gates (per blind-rotate kernel launch, averaged over the run's launches)"
for parts in 1 2 3 4; do
  for slice in 0 768; do
    for grp in "FETCH_SIZE GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum"; do
      d="$O/p${parts}_s${slice}_$(echo $grp | cut -d' ' -f1)"
      EOC_TFHE_BR_PARTS=$parts EOC_TFHE_BR_SLICE=$slice rocprofv3 --pmc $grp --output-format csv -d "$d" -- python3 bench.py --pset B --gates 1024 --steps 3 --warmup 1 --no-cpu-baseline --no-secondary --no-host-legs > /dev/null 2> "$d.err" || echo "pass failed: $d"
    done
    python3 - "$O" $parts $slice <<'PY'
import csv, glob, sys
O, parts, sl = sys.argv[1], sys.argv[2], sys.argv[3]
tot = {}
n = 0
for grp in ("FETCH_SIZE", "TCC_HIT_sum"):
    for f in glob.glob(f"{O}/p{parts}_s{sl}_{grp}/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if "k_blind_rotate" not in row.get("Kernel_Name", ""):
                continue
            tot[row["Counter_Name"]] = tot.get(row["Counter_Name"], 0.0) + float(row["Counter_Value"])
            if row["Counter_Name"] == "FETCH_SIZE":
                n += 1
hit, miss = tot.get("TCC_HIT_sum", 0), tot.get("TCC_MISS_sum", 0)
print(f"parts={parts} slice={sl}: launches {n}  FETCH_SIZE/launch {tot.get('FETCH_SIZE', 0) / max(1, n):.0f} (counter units)  "
      f"TCC hit rate {hit / max(1.0, hit + miss):.4f}")
PY
  done
done
