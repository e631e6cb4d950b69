#!/bin/bash
# Build timing variants of the library HERE (hipcc cross-compiles; no GPU minutes spent on compiling) and
# write a runner for the GPU box.  Usage:  bash tools/variants.sh "VARIANT1" "VARIANT2" ...
#   a variant is a space-separated list of compiler flags and/or `patch:NAME` items; patch:NAME applies
#   tools/patches/NAME.patch to a scratch copy of eoc_tfhe_amd/csrc (the shipped source carries no ablation or
#   tuning switches: measured-and-rejected forms live as patches, their results in DESIGN.md)
#   then:  gpurun -- 'bash eoc_tfhe_amd/_build/run_variants.sh'      (BENCH_ARGS env is passed to bench.py)
# Variant 0 is always the shipped source with the default flags.  Ablation patches give wrong results: timing only.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
B=$ROOT/eoc_tfhe_amd/_build
mkdir -p "$B"
python -m eoc_tfhe_amd.build >/dev/null 2>&1 || true
i=0
: > "$B/variants.txt"
pids=()
for spec in "" "$@"; do
  (
    src=$ROOT/eoc_tfhe_amd/csrc
    flags=""
    for item in $spec; do
      case "$item" in
        patch:*)
          if [ "$src" = "$ROOT/eoc_tfhe_amd/csrc" ]; then
            rm -rf "$B/var_src_$i" && mkdir -p "$B/var_src_$i/eoc_tfhe_amd" "$B/var_src_$i/include"
            cp -r "$ROOT/eoc_tfhe_amd/csrc" "$B/var_src_$i/eoc_tfhe_amd/csrc"
            cp "$ROOT"/include/*.h "$B/var_src_$i/include/"
            src=$B/var_src_$i/eoc_tfhe_amd/csrc
          fi
          patch -s -p1 -d "$B/var_src_$i" < "$ROOT/tools/patches/${item#patch:}.patch" ;;
        *) flags="$flags $item" ;;
      esac
    done
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result -w $flags \
       -c "$src/engine.hip" -o "$B/var_engine_$i.o" &&
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 "$B/var_engine_$i.o" "$B/multi.o" "$B/host.o" "$B/legacy.o" -o "$B/var_$i.so" -lgomp -ldl -Wl,-rpath,/opt/rocm/lib
  ) &
  pids+=($!)
  echo "$i|$spec" >> "$B/variants.txt"
  i=$((i+1))
  if (( ${#pids[@]} >= 4 )); then wait "${pids[0]}"; pids=("${pids[@]:1}"); fi
done
wait
cat > "$B/run_variants.sh" <<'EOS'
#!/bin/bash
cd "$GRAFT_REPO_ROOT"
while IFS='|' read -r i flags; do
  res=$(EOC_TFHE_LIB=$PWD/eoc_tfhe_amd/_build/var_$i.so python bench.py --steps ${STEPS:-10} --warmup 3 --no-cpu-baseline --no-secondary ${BENCH_ARGS} 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['kernels_ms'], d['value'], 'ok' if d['decrypt_ok'] else 'WRONG-RESULT')")
  echo "[$i] variant='$flags' -> $res"
done < eoc_tfhe_amd/_build/variants.txt
EOS
echo "built $i variants"
