#!/bin/bash
set -u
cd /root/repo
mkdir -p gpurun_out/r02d
timeout -k 10 600 python -m pytest tests -m gpu -x -q > gpurun_out/r02d/pytest.log 2>&1; tail -3 gpurun_out/r02d/pytest.log
for pose in 0 1 2; do
python tools/perf_probe.py --scene monu9 --pose $pose --variants 1 --cull 0,1,2 --reps 30 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('monu9 pose $pose cull', d['cull'], d['ms_med'], d['ms_min'], d['mrays_s'], d['same_as_first'])"
done
python tools/perf_probe.py --scene config3a --variants 1 --cull 0,1,2 --reps 30 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('config3a cull', d['cull'], d['ms_med'], d['ms_min'], d['mrays_s'], d['same_as_first'])"
python tools/perf_probe.py --variants 1 --cull 0,2 --reps 30 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('terrain cull', d['cull'], d['ms_med'], d['ms_min'], d['mrays_s'], d['same_as_first'])"
