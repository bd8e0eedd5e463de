#!/bin/bash
cd /root/repo
python tools/perf_probe.py --variants 1 --grid 1536,1280,1024,768,512 --reps 30 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('terrain grid', d['grid'], d['ms_med'], d['ms_min'], d['mrays_s'], d['same_as_first'])"
python tools/perf_probe.py --w 3840 --h 2160 --variants 1 --grid 1536,1280,1024,768 --reps 12 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('terrain4k grid', d['grid'], d['ms_med'], d['ms_min'], d['mrays_s'], d['same_as_first'])"
