#!/bin/bash
# wave-priority duty (EOC_TFHE_PRIO_DUTY, sixteenths; -1 = off) against launch size (GPU box)
cd "$GRAFT_REPO_ROOT"
for g in ${GATES:-1024 4096 16384}; do
  for d in ${DUTIES:--1 0 8 12 16}; do
    EOC_TFHE_PRIO_DUTY=$d python bench.py --gates $g --steps ${STEPS:-6} --warmup 2 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']
print(f'gates=$g duty=$d BR={k[\"blind_rotate\"]:8.4f} ms  gates/s={d[\"value\"]:9.0f} ok={d[\"decrypt_ok\"]}')"
  done
done
