#!/bin/bash
# lab: per-kernel times of cfg4 (one solve alone) for the product library and for each variant under lab/variants/
#   lab/run_variants.sh <out-file> <script + args> -- <variant names...>   (run on the GPU box)
# A variant is loaded through QMCP_HIP_LIB (genome-downsampler_amd/__init__.py): the product library is never overwritten.
OUT="$1"; shift
CMD=(); while [ "$1" != "--" ]; do CMD+=("$1"); shift; done; shift
for v in product "$@"; do
    echo "=== $v" >> "$OUT"
    if [ "$v" = product ]; then python "${CMD[@]}" >> "$OUT" 2>&1
    else QMCP_HIP_LIB="$PWD/lab/variants/$v/libqmcp_hip.so" python "${CMD[@]}" >> "$OUT" 2>&1; fi
done
