#!/bin/bash
# same-box A/B of library builds on the benchmark frame, round-robin, with a summary (round 4: tools/ab.sh's two passes of 40 frames
# leave +-1.5 % of noise, more than most single changes are worth):
#   ROUNDS=4 REPS=150 tools/ab2.sh build/a.so build/b.so ...      (AB_ARGS: extra perf_probe.py arguments)
ROUNDS=${ROUNDS:-4}; REPS=${REPS:-150}
for rep in $(seq $ROUNDS); do
for entry in "$@"; do
  lib=${entry%%:*}; envs=""; [ "$entry" != "$lib" ] && envs=${entry#*:}   # build.so or build.so:VAR=value (an environment switch of the library)
  env $envs SVO_HIP_LIB=$PWD/$lib python tools/perf_probe.py --lod 1500 --variants 1 --refill ${REFILL:-32} --schedule 2 --reps $REPS ${AB_ARGS:-} 2>/dev/null | grep '^{' | python3 -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('$entry', d['ms_med'], d['ms_min'], d['sig'])"
done; done | tee /tmp/ab2_raw.txt | python3 -c "
import sys,collections
r=collections.defaultdict(list); sig={}
for l in sys.stdin:
    p=l.split(); r[p[0]].append((float(p[1]),float(p[2]))); sig[p[0]]=p[3]
base=None
for k,v in r.items():
    med=sorted(x[0] for x in v); m=sum(med)/len(med)
    if base is None: base=m
    print(f'{k:28s} mean of medians {m:.4f} ms  (min {med[0]:.4f} max {med[-1]:.4f}; fastest frame {min(x[1] for x in v):.4f})  {100*(m/base-1):+.1f} %  sig {sig[k]}')
"
