#!/bin/bash
# A/B on ONE box: the rotation amounts read back through LDS (shipped) against scalar loads (EOC_TFHE_SCALAR_ABAR=1),
# headline (1024 NAND, Set A), wide (16 384 NAND), Set B; two alternating passes.  Usage: gpurun -- 'bash tools/ab_abar.sh'
cd "${GRAFT_REPO_ROOT:-.}"
for pass in 1 2; do
  for form in 0 1; do
    for args in "--gates 1024 --steps 40" "--gates 16384 --steps 6" "--gates 1024 --steps 30 --pset B"; do
      res=$(EOC_TFHE_SCALAR_ABAR=$form python bench.py --warmup 3 --no-cpu-baseline --no-secondary --no-host-legs $args 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms']['blind_rotate'], d['value'], d['clock']['sclk_mhz_under_load'], d['clock']['package_power_w_max_seen'], 'ok' if d['decrypt_ok'] else 'WRONG-RESULT')")
      echo "[scalar=$form] $args -> $res"
    done
  done
done
