#!/bin/bash
cd /root/repo
for sc in monu9 config3b terrain; do
python tools/perf_probe.py --scene $sc --variants 1 --refill 8,16,24,32,48,64 --reps 30 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$sc refill', d['refill'], d['ms_med'], d['ms_min'], d['mrays_s'], d['same_as_first'])"
done
