#!/bin/bash
set -o pipefail
O=gpurun_out/r05c14; mkdir -p $O
echo "== refill_min sweep 1080p"
for r in 1 2 3; do timeout -k 10 200 python tools/perf_probe.py --lod 1500 --variants 1 --refill 16,24,32,40,48,56 --schedule 2 --reps 100 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('1080p refill', d['refill'], d['ms_med'], d['ms_min'])"; done | tee $O/refill_sweep_1080p.log
echo "== refill_min sweep 4k"
for r in 1 2; do timeout -k 10 300 python tools/perf_probe.py --lod 1500 --w 3840 --h 2160 --variants 1 --refill 16,24,32,40,48 --schedule 2 --reps 40 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('4k refill', d['refill'], d['ms_med'], d['ms_min'])"; done | tee $O/refill_sweep_4k.log
