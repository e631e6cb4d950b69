#!/bin/bash
# Build diagnostic variants of the library (wrong results, timing only) and time k_blind_rotate in each.
# Usage on the GPU box: bash tools/ablate.sh "FLAGSET1" "FLAGSET2" ...   (each a space-separated list of -D flags)
set -e
cd "$GRAFT_REPO_ROOT"
mkdir -p gpurun_out/abl
i=0
for flags in "" "$@"; do
  out=gpurun_out/abl/lib_$i.so
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result $flags \
     -c eoc_tfhe_amd/csrc/engine.hip -o gpurun_out/abl/engine_$i.o
  /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 gpurun_out/abl/engine_$i.o eoc_tfhe_amd/_build/multi.o eoc_tfhe_amd/_build/host.o eoc_tfhe_amd/_build/legacy.o -o $out -lgomp -ldl -Wl,-rpath,/opt/rocm/lib
  res=$(EOC_TFHE_LIB=$PWD/$out python bench.py --steps 6 --warmup 2 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['value'], 'ok' if d['decrypt_ok'] else 'WRONG-RESULT(expected for ablations)')")
  echo "[$i] flags='$flags' -> $res"
  i=$((i+1))
done
