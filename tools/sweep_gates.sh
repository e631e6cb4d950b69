#!/bin/bash
# blind-rotate / key-switch time against the launch size (GPU box).  Usage: bash tools/sweep_gates.sh 512 1024 ...
cd "$GRAFT_REPO_ROOT"
for g in "$@"; do
  python bench.py --gates $g --steps 8 --warmup 2 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); k=d['kernels_ms']; g=$g
print(f'gates={g:6d} BR={k[\"blind_rotate\"]:8.4f} ms  KS={k[\"keyswitch\"]:7.4f} ms  BR per 1024 = {k[\"blind_rotate\"]*1024/g:6.3f}  gates/s={d[\"value\"]:9.0f}')"
done
