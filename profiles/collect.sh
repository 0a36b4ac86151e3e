#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/stats/...      rocprofv3 --kernel-trace --stats of the default bench command
#   gpurun_out/prof/pmc_*/...      one --pmc pass per counter (FETCH_SIZE, WRITE_SIZE), as the guide prescribes
#   gpurun_out/bench_*.json        bench lines (cfg4 default, cfg2, and the one printed under rocprofv3)
# Copy what is to be judged into profiles/ afterwards.
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT/prof"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 10 --warmup 2 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
python3 "$ROOT/bench.py" --workload cfg2 --steps 50 --warmup 5 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof/stats" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof/stats.err"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/prof/pmc_$c" -o pmc -- python3 "$ROOT/bench.py" --steps 2 --warmup 1 --in-flight 1 --no-extras \
        > "$OUT/prof/pmc_$c.json" 2> "$OUT/prof/pmc_$c.err"
done
python3 "$ROOT/profiles/summarize_pmc.py" "$OUT/prof" "$OUT/prof/pmc_traffic_cfg4.json" \
    "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 2 --warmup 1 --in-flight 1, profiles/collect.sh, $(date -u +%Y-%m-%d)" \
    > "$OUT/prof/pmc_summary.txt"
find "$OUT/prof" -name "*.csv" | head -20
