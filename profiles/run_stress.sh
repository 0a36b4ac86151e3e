#!/bin/bash
# The stress harnesses on the code as it stands, every log headed by the commit it ran on (run on the GPU box from the
# repo root: profiles/run_stress.sh <commit hash> [scale]; the hash is passed in because the box has no .git).
H="$1"; S="${2:-1}"
run() { out="gpurun_out/r04_stress_$1.log"; shift; echo "commit $H; command: $*" > "$out"; timeout -k 10 1500 "$@" >> "$out" 2>&1 || echo "FAILED rc=$?" >> "$out"; tail -1 "$out"; }
run pm python lab/stress_pm.py 0 $((600 * S))
run more python lab/stress_more.py 0 $((300 * S))
run ev python lab/stress_ev.py 0 $((500 * S))
run near_uniform python lab/stress_near_uniform.py $((1000 * S)) 0
run near_uniform_shallow python lab/stress_near_uniform.py $((250 * S)) 100000 shallow
run spec python lab/stress_spec.py $((300 * S))
