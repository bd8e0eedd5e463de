#!/bin/bash
# same-box A/B of library builds in the counting (adaptive) mode: tools/ab_count.sh build_ab/a.so build_ab/b.so ...
# per build and repetition: tools/count_probe.py (primary rays, counters cleared / carried over) and tools/default_mode_probe.py
for rep in 1 2 3; do
for lib in "$@"; do
  c=$(SVO_HIP_LIB=$PWD/$lib python tools/count_probe.py --reps 40 2>/dev/null | grep -o 'median [0-9.]*' | tr '\n' ' ')
  d=$(SVO_HIP_LIB=$PWD/$lib python tools/default_mode_probe.py --frames 60 2>/dev/null | grep -o 'shade [0-9.]* ms wall' )
  echo "$lib rep $rep: static/cleared/carried $c | default mode $d"
done; done
