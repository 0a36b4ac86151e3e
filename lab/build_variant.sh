#!/bin/bash
# lab: build a variant of libqmcp_hip.so with extra compiler flags into lab/variants/<name>/ (not part of the product)
#   lab/build_variant.sh <name> "<extra hipcc flags>" [git-ref to build the kernels from instead of the working tree]
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"
NAME="$1"; FLAGS="$2"; REF="$3"
OUT="$ROOT/lab/variants/$NAME"; mkdir -p "$OUT"
SRC="$ROOT"
if [ -n "$REF" ]; then
    SRC="$(mktemp -d)"; git -C "$ROOT" archive "$REF" genome-downsampler_amd/csrc include | tar -x -C "$SRC"
fi
HF="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -I$SRC/include -I$SRC/genome-downsampler_amd/csrc -Wall -Wno-unused-function $FLAGS"
/opt/rocm/bin/hipcc $HF -c "$SRC/genome-downsampler_amd/csrc/qmcp_kernels.hip" -o "$OUT/qmcp_kernels.o" &
/opt/rocm/bin/hipcc $HF -c "$SRC/genome-downsampler_amd/csrc/qmcp_api.hip" -o "$OUT/qmcp_api.o" &
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC "$OUT/qmcp_kernels.o" "$OUT/qmcp_api.o" -o "$OUT/libqmcp_hip.so"
rm -f "$OUT"/*.o
echo "built $OUT/libqmcp_hip.so"
