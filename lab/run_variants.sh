#!/bin/bash
# lab: per-kernel times of cfg4 (one solve alone) for the product library and for each variant under lab/variants/
#   lab/run_variants.sh <out-file> <script + args> -- <variant names...>   (run on the GPU box)
OUT="$1"; shift
CMD=(); while [ "$1" != "--" ]; do CMD+=("$1"); shift; done; shift
LIB=genome-downsampler_amd/lib/libqmcp_hip.so
cp $LIB /tmp/product.so
for v in product "$@"; do
    if [ "$v" = product ]; then cp /tmp/product.so $LIB; else cp lab/variants/$v/libqmcp_hip.so $LIB; fi
    echo "=== $v" >> "$OUT"
    python "${CMD[@]}" >> "$OUT" 2>&1
done
cp /tmp/product.so $LIB
