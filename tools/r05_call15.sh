#!/bin/bash
set -o pipefail
O=gpurun_out/r05c15; mkdir -p $O
echo "== block shape sweep at refill 32 (2: 4x16, 3: 8x8, 4: 16x4)"
for r in 1 2; do timeout -k 10 200 python tools/perf_probe.py --lod 1500 --variants 1 --refill 32 --block 2,3,4 --schedule 2 --reps 100 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('1080p block', d['block'], d['ms_med'], d['ms_min'])"; done | tee $O/block_sweep_1080p.log
echo "== bench.py"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo rc $?; python tools/show_bench.py $O/bench.json 2>/dev/null | head -60 || head -c 3000 $O/bench.json
