#!/bin/bash
# AddressSanitizer + UBSan over everything that runs on the CPU (the GPU pool has no sanitizer runs): the host mirror
# (BAM ingest / emit, containers, amplicon files, the plugin adapter) under tests/test_bam_io.py and
# tests/test_abi_and_host.py, and the oracle under its golden / selection / model tests.  Builds into /tmp/qmcp_asan;
# the product libraries are not touched (QMCP_HOST_LIB / QMCP_ORACLE_LIB point the loaders at the sanitized builds).
#   tools/sanitize_cpu.sh        (from the repo root, after `make`)
set -e
ROOT="$(cd "$(dirname "$0")/.." && pwd)"; cd "$ROOT"
OUT=/tmp/qmcp_asan; mkdir -p $OUT
PKG=genome-downsampler_amd
SAN="-fsanitize=address,undefined -fno-omit-frame-pointer -O1 -g"
g++ $SAN -std=c++17 -fPIC -ffp-contract=off -Wall -Iinclude -I$PKG/host/include -DHIP_ENABLED -shared \
    $PKG/host/src/bam_api.cpp $PKG/host/src/reads_gen.cpp $PKG/host/src/quasi_mcp_hip_solver.cpp $PKG/host/src/amplicon_set.cpp \
    $PKG/host/src/bam_io.cpp $PKG/host/src/host_c_api.cpp -L$PKG/lib -lqmcp_hip -lz -lpthread -Wl,-rpath,$ROOT/$PKG/lib -o $OUT/libqmcp_host.so
gcc $SAN -std=c11 -fPIC -Wall -shared oracle/qmcp_oracle.c -o $OUT/libqmcp_oracle.so
export LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)"
export ASAN_OPTIONS=detect_leaks=0:halt_on_error=1 UBSAN_OPTIONS=print_stacktrace=1
export QMCP_HOST_LIB=$OUT/libqmcp_host.so QMCP_ORACLE_LIB=$OUT/libqmcp_oracle.so
python -m pytest tests/test_bam_io.py tests/test_abi_and_host.py tests/test_oracle_golden.py tests/test_oracle_selection.py \
    tests/test_oracle_forgetting.py tests/test_near_uniform_model.py -q -s -m "not gpu" > $OUT/run.log 2>&1 || { tail -30 $OUT/run.log; exit 1; }
n=$(grep -c "runtime error\|AddressSanitizer" $OUT/run.log || true)
tail -1 $OUT/run.log
echo "sanitizer reports: $n"
test "$n" = 0
