#!/bin/bash
# same-box A/B of library builds on the benchmark frame: tools/ab.sh build_ab/a.so build_ab/b.so ...
for rep in 1 2; do
for lib in "$@"; do
  SVO_HIP_LIB=$PWD/$lib python tools/perf_probe.py --lod 1500 --variants 1 --refill ${REFILL:-32} --schedule 2 --reps 40 ${AB_ARGS:-} 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$lib', d['variant'], d['ms_med'], d['ms_min'], d['mrays_s'], d['sig'])"
done; done
