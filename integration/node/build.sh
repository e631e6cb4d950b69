#!/bin/bash
# Build the N-API addon next to libeoc_tfhe_gpu.so (no node-gyp needed: one C file, N-API is ABI-stable).
set -e
HERE=$(cd "$(dirname "$0")" && pwd); ROOT=$(cd "$HERE/../.." && pwd)
INC=${NODE_INCLUDE:-/usr/include/node}
gcc -std=gnu11 -O2 -fPIC -shared -Wall -I"$INC" -I"$ROOT/include" "$HERE/eoc_tfhe_node.c" -o "$HERE/eoc_tfhe.node" \
    -L"$ROOT/eoc_tfhe_amd" -leoc_tfhe_gpu -Wl,-rpath,"$ROOT/eoc_tfhe_amd" -Wl,-rpath,/opt/rocm/lib
echo "$HERE/eoc_tfhe.node"
