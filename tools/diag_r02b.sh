#!/bin/bash
set -u
cd /root/repo
mkdir -p gpurun_out/r02b
timeout -k 10 900 python -m pytest tests -m gpu -x -q > gpurun_out/r02b/pytest.log 2>&1 || { tail -30 gpurun_out/r02b/pytest.log; exit 1; }
tail -3 gpurun_out/r02b/pytest.log
for pt in 0 1; do
  timeout -k 10 300 python bench.py --steps 100 --warmup 10 --no-cpu-baseline --opt PAIR_TABLE=$pt > gpurun_out/r02b/bench_pairs$pt.json 2> gpurun_out/r02b/bench_pairs$pt.err || { tail -5 gpurun_out/r02b/bench_pairs$pt.err; exit 1; }
  python3 -c "
import json;d=json.load(open('gpurun_out/r02b/bench_pairs$pt.json'));print('pairs $pt', d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['frac'], d['config'].get('cold_frame_ms'), d['config'].get('motion_ms'), d['config']['also'])"
done
for s in terrain config2 config3b; do
  timeout -k 10 300 python tools/wave_timeline.py --scene $s --json gpurun_out/r02b/timeline_$s.json > gpurun_out/r02b/timeline_$s.log 2>&1 || { echo "timeline $s failed"; tail -5 gpurun_out/r02b/timeline_$s.log; exit 1; }
done
echo done
