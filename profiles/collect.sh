#!/bin/bash
# Collects the round's profile artefacts on the GPU box (run through gpurun from the repo root):
#   gpurun_out/prof/stats/...         rocprofv3 --kernel-trace --stats of the bench command, timed configuration only
#   gpurun_out/prof/stats_extras/...  the same with the default line's other shapes (other_configs)
#   gpurun_out/prof/calib_*/...       one --pmc pass per counter over lab/pmc_calib (kernels of known byte counts)
#   gpurun_out/prof/d<k>_pmc_*/...    one --pmc pass per counter (FETCH_SIZE, WRITE_SIZE), as the guide prescribes, for
#                                     k = 1 and 2 solves in flight (the timed configuration is k = 2)
#   gpurun_out/bench_*.json           bench lines (cfg4 default with 200 steps, cfg2, and the one printed under rocprofv3)
# Copy what is to be judged into profiles/ afterwards.
set -eo pipefail
ROOT="${GRAFT_REPO_ROOT:-$(pwd)}"
OUT="$ROOT/gpurun_out"
mkdir -p "$OUT/prof"
cd /tmp && export TMPDIR=/tmp
python3 "$ROOT/bench.py" --steps 200 --warmup 5 > "$OUT/bench_cfg4.json" 2> "$OUT/bench_cfg4.err"
echo "bench cfg4 done"
python3 "$ROOT/bench.py" --workload cfg2 --steps 50 --warmup 5 > "$OUT/bench_cfg2.json" 2> "$OUT/bench_cfg2.err"
# the timed configuration alone (--no-extras: the other shapes of the default line launch the same kernels on other
# sizes -- the near-uniform route re-sweeps parts of contigs with k_sweep_uniform_ev -- and would blur the averages
# the roofline is checked against); the line with the extras is profiled separately
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof/stats" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 --no-extras \
    > "$OUT/bench_under_rocprof.json" 2> "$OUT/prof/stats.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/prof/stats_extras" -o run -- python3 "$ROOT/bench.py" --steps 5 --warmup 2 \
    > "$OUT/bench_under_rocprof_with_extras.json" 2> "$OUT/prof/stats_extras.err"
echo "kernel stats done"
for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c --output-format csv -d "$OUT/prof/calib_$c" -o pmc -- "$ROOT/lab/pmc_calib" > "$OUT/prof/calib_$c.log" 2>&1
done
python3 "$ROOT/profiles/summarize_pmc.py" calibrate "$OUT/prof" "$OUT/prof/pmc_calibration.json" > "$OUT/prof/pmc_calibration.txt"
echo "calibration done"
for depth in 1 2; do
    for c in FETCH_SIZE WRITE_SIZE; do
        rocprofv3 --pmc $c --output-format csv -d "$OUT/prof/d${depth}_pmc_$c" -o pmc -- python3 "$ROOT/bench.py" --steps 4 --warmup 1 --in-flight $depth --no-extras \
            > "$OUT/prof/d${depth}_pmc_$c.json" 2> "$OUT/prof/d${depth}_pmc_$c.err"
    done
    python3 "$ROOT/profiles/summarize_pmc.py" summarize "$OUT/prof" "d${depth}_" "$OUT/prof/pmc_traffic_cfg4_inflight${depth}.json" "$OUT/prof/pmc_calibration.json" \
        "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py --steps 4 --warmup 1 --in-flight $depth --no-extras, profiles/collect.sh, $(date -u +%Y-%m-%d)" \
        > "$OUT/prof/pmc_summary_inflight${depth}.txt"
    echo "pmc depth $depth done"
done
find "$OUT/prof" -name "*.csv" | head -30
