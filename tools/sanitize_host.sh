#!/bin/bash
# Host-side AddressSanitizer + UBSan build of the whole library (device code is NOT instrumented: GPU ASan is
# unavailable on the pool) linked into the plain-C host tests/c/abi_smoke.c.  Usage: tools/sanitize_host.sh <outdir>
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=${1:-/tmp/eoc_san}; mkdir -p "$OUT"
CL=/opt/rocm/lib/llvm/bin
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -g -O1"
/opt/rocm/bin/hipcc $SAN -fno-gpu-sanitize -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result \
    -c "$ROOT/eoc_tfhe_amd/csrc/engine.hip" -o "$OUT/engine.o"
/opt/rocm/bin/hipcc $SAN -fno-gpu-sanitize -std=c++17 -fPIC -ffp-contract=off --offload-arch=gfx950 -Wno-unused-value -Wno-unused-result \
    -c "$ROOT/eoc_tfhe_amd/csrc/multi.hip" -o "$OUT/multi.o"
for f in host legacy; do
  $CL/clang++ $SAN -std=c++17 -fPIC -ffp-contract=off -fopenmp -mavx2 -mfma -c "$ROOT/eoc_tfhe_amd/csrc/$f.cpp" -o "$OUT/$f.o"
done
$CL/clang $SAN -std=c11 -I"$ROOT/include" -c "$ROOT/tests/c/abi_smoke.c" -o "$OUT/smoke.o"
/opt/rocm/bin/hipcc $SAN -fno-gpu-sanitize --offload-arch=gfx950 "$OUT/engine.o" "$OUT/multi.o" "$OUT/host.o" "$OUT/legacy.o" "$OUT/smoke.o" -o "$OUT/abi_smoke_san" \
    -fopenmp -ldl -Wl,-rpath,/opt/rocm/lib -Wl,-rpath,/opt/rocm/lib/llvm/lib
echo "$OUT/abi_smoke_san"
