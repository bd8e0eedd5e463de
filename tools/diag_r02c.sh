#!/bin/bash
set -u
cd /root/repo
mkdir -p gpurun_out/r02c
bash tools/ab.sh build_ab/r02_base.so build_ab/r02_reorder.so > gpurun_out/r02c/ab_terrain.log 2>&1
cat gpurun_out/r02c/ab_terrain.log
AB_ARGS="--scene monu9" bash tools/ab.sh build_ab/r02_base.so build_ab/r02_reorder.so > gpurun_out/r02c/ab_monu9.log 2>&1
cat gpurun_out/r02c/ab_monu9.log
AB_ARGS="--scene config3b" bash tools/ab.sh build_ab/r02_base.so build_ab/r02_reorder.so > gpurun_out/r02c/ab_c3b.log 2>&1
cat gpurun_out/r02c/ab_c3b.log
AB_ARGS="--pairs 1" bash tools/ab.sh build_ab/r02_reorder.so > gpurun_out/r02c/ab_pairs.log 2>&1
cat gpurun_out/r02c/ab_pairs.log
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02c/pytest.log 2>&1; tail -3 gpurun_out/r02c/pytest.log
for s in terrain config2; do
  timeout -k 10 300 python tools/wave_timeline.py --scene $s --opt PAIR_TABLE=0 --json gpurun_out/r02c/timeline_$s.json > gpurun_out/r02c/timeline_$s.log 2>&1 || { echo "timeline $s failed"; tail -5 gpurun_out/r02c/timeline_$s.log; }
done
