#!/bin/bash
set -o pipefail
O=gpurun_out/r05c13; mkdir -p $O
echo "== ab 1080p cost class width (2 / 4 / 8 steps)"; ROUNDS=3 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_cur.so build_ab/r05_cs1.so build_ab/r05_cs3.so 2>&1 | tee $O/ab_cs_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=2 REPS=40 timeout -k 10 600 tools/ab2.sh build_ab/r05_cur.so build_ab/r05_cs1.so 2>&1 | tee $O/ab_cs_4k.log
echo "== refill_min sweep (current build)"
for r in 1 2; do timeout -k 10 200 python tools/perf_probe.py --lod 1500 --variants 1 --refill 8,12,16,20,24,32 --schedule 2 --reps 100 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('refill', d['refill'], d['ms_med'], d['ms_min'])"; done | tee $O/refill_sweep.log
