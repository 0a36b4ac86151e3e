#!/bin/bash
# rocprofv3 kernel statistics of the other measured shapes: cfg3 at full size, cfg5's
# shape at 1/64 scale, shallow data split at cut points (uniform and mixed spans), cfg3's shape with a clipped tail of read
# lengths (mixed-span route), one GPU's share of cfg5 (1/8 scale, no oracle).  Run through gpurun
# from the repo root; copy gpurun_out/prof_other/*/run_kernel_stats.csv into profiles/ afterwards.
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out/prof_other"
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg3" -o run -- python3 "$ROOT/lab/prof_cfg3.py" > "$OUT/cfg3.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5shape" -o run -- python3 "$ROOT/lab/prof_cfg5_shape.py" > "$OUT/cfg5shape.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/shallow" -o run -- python3 "$ROOT/lab/prof_cut_segments.py" 2e7 30 1.0 > "$OUT/shallow.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/shallow_mixed" -o run -- python3 "$ROOT/lab/prof_cut_segments.py" 2e7 30 1.0 0 100 > "$OUT/shallow_mixed.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mixed_amplicon" -o run -- python3 "$ROOT/lab/prof_mixed_amplicon.py" > "$OUT/mixed_amplicon.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/cfg5share" -o run -- python3 "$ROOT/lab/check_cfg5_share.py" 0.125 0 > "$OUT/cfg5share.log" 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/mixed_wgs" -o run -- python3 "$ROOT/lab/prof_mixed_wgs.py" > "$OUT/mixed_wgs.log" 2>&1
find "$OUT" -name "run_kernel_stats.csv"
