for r in 1 2; do for B in 0 8 10 12; do python tools/perf_probe.py --lod 1500 --variants 1 --refill 32 --schedule 2 --reps 100 --relayout $B 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('1080p relayout $B', d['ms_med'], d['ms_min'])"; done; done
for B in 0 10; do python tools/perf_probe.py --lod 1500 --w 3840 --h 2160 --variants 1 --refill 32 --schedule 2 --reps 40 --relayout $B 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('4k relayout $B', d['ms_med'], d['ms_min'])"; done
