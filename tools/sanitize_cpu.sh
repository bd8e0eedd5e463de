#!/bin/bash
# CPU sanitizer pass (GPU ASan is not available on this pool): builds the host half of the drop-in
# (csrc/svo_host.cpp: octree / .vox / .rsvo / world / adaptive list processing) and the oracle with
# gcc's AddressSanitizer + UBSan, links the host objects into an otherwise normal libsvo_hip.so, and
# runs the CPU test suite against both.  Everything is written under /tmp/svo_asan.
# usage: tools/sanitize_cpu.sh [pytest args]
set -euo pipefail
ROOT=$(cd "$(dirname "$0")/.." && pwd)
OUT=/tmp/svo_asan
mkdir -p "$OUT"
SAN="-fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -g -O1"
g++ $SAN -std=c++17 -ffp-contract=off -fPIC -I"$ROOT/include" -c "$ROOT/octree-tracer_amd/csrc/svo_host.cpp" -o "$OUT/svo_host.o"
HIPFLAGS="-O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fPIC -I$ROOT/include -Wno-unused-function"
/opt/rocm/bin/hipcc $HIPFLAGS -c "$ROOT/octree-tracer_amd/csrc/svo_kernels.hip" -o "$OUT/svo_kernels.o"
/opt/rocm/bin/hipcc $HIPFLAGS -c "$ROOT/octree-tracer_amd/csrc/svo_abi.cpp" -o "$OUT/svo_abi.o"
/opt/rocm/bin/hipcc $HIPFLAGS -c "$ROOT/octree-tracer_amd/csrc/svo_comm.cpp" -o "$OUT/svo_comm.o"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -fPIC -shared -o "$OUT/libsvo_hip.so" "$OUT/svo_kernels.o" "$OUT/svo_abi.o" "$OUT/svo_comm.o" "$OUT/svo_host.o" -ldl
gcc $SAN -std=c11 -ffp-contract=off -fno-fast-math -fPIC -pthread -shared -o "$OUT/libsvo_oracle.so" "$ROOT/oracle/svo_oracle.c" -lm -lpthread
cd "$ROOT"
LD_PRELOAD="$(gcc -print-file-name=libasan.so):$(gcc -print-file-name=libubsan.so)" \
ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
SVO_HIP_LIB="$OUT/libsvo_hip.so" SVO_ORACLE_LIB="$OUT/libsvo_oracle.so" \
    python -m pytest tests -q -m "not gpu" -p no:cacheprovider --deselect tests/test_cpp_host.py "$@"   # that test links a plain C++ program against the library
