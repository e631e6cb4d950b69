#!/bin/bash
# Build timing variants of the library HERE (hipcc cross-compiles; no GPU minutes spent on compiling) and
# write a runner for the GPU box.  Usage:  bash tools/variants.sh "FLAGSET1" "FLAGSET2" ...
#   then:  gpurun -- 'bash eoc_tfhe_amd/_build/run_variants.sh'      (BENCH_ARGS env is passed to bench.py)
# Variant 0 is always the default build flags.  Results may be wrong for ablation flags: timing only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/eoc_tfhe_amd/_build
mkdir -p "$B"
python -m eoc_tfhe_amd.build >/dev/null 2>&1 || true
i=0
: > "$B/variants.txt"
pids=()
for flags in "" "$@"; do
  (
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -w $flags \
       -c "$ROOT/eoc_tfhe_amd/csrc/engine.hip" -o "$B/var_engine_$i.o" &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$B/var_engine_$i.o" "$B/multi.o" "$B/host.o" "$B/legacy.o" -o "$B/var_$i.so" -lgomp -ldl -Wl,-rpath,/opt/rocm/lib
  ) &
  pids+=($!)
  echo "$i|$flags" >> "$B/variants.txt"
  i=$((i+1))
  if (( ${#pids[@]} >= 4 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
cat > "$B/run_variants.sh" <<'EOS'
#!/bin/bash
cd "$GRAFT_REPO_ROOT"
while IFS='|' read -r i flags; do
  res=$(EOC_TFHE_LIB=$PWD/eoc_tfhe_amd/_build/var_$i.so python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['value'], 'ok' if d['decrypt_ok'] else 'WRONG-RESULT')")
  echo "[$i] flags='$flags' -> $res"
done < eoc_tfhe_amd/_build/variants.txt
EOS
echo "built $i variants"
