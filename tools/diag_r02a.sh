#!/bin/bash
# round-2 diagnostics, one gpurun call: VALU issue rate, per-phase timelines of four scenes
set -u
cd /root/repo
mkdir -p gpurun_out/r02a
timeout -k 5 60 ./build_ab/valu_rate > gpurun_out/r02a/valu_rate.log 2>&1 || { echo "valu_rate failed"; exit 1; }
for s in terrain config2 config3a config3b; do
  timeout -k 10 300 python tools/wave_timeline.py --scene $s --json gpurun_out/r02a/timeline_$s.json > gpurun_out/r02a/timeline_$s.log 2>&1 || { echo "timeline $s failed"; tail -5 gpurun_out/r02a/timeline_$s.log; exit 1; }
  echo "timeline $s ok"
done
timeout -k 10 300 python tools/wave_timeline.py --scene terrain --w 3840 --h 2160 --json gpurun_out/r02a/timeline_terrain4k.json > gpurun_out/r02a/timeline_terrain4k.log 2>&1 || exit 1
timeout -k 10 300 python bench.py --steps 50 --warmup 5 --no-cpu-baseline > gpurun_out/r02a/bench.json 2> gpurun_out/r02a/bench.err || exit 1
echo done
