#!/bin/bash
set -o pipefail
O=gpurun_out/r05c21; mkdir -p $O
echo "== grid size sweep 1080p (32400 strips: 1620 workgroups = 5.0 strips per wave, 1792 = 4.52)"
for r in 1 2 3; do timeout -k 10 300 python tools/perf_probe.py --lod 1500 --variants 1 --refill 32 --grid 0,1792,1720,1660,1620,1580,1540,1350 --schedule 2 --reps 80 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('1080p grid', d['grid'], d['ms_med'], d['ms_min'])"; done | tee $O/grid_sweep_1080p.log
