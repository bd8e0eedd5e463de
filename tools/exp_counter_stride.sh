#!/bin/bash
# round 5: the 64 claim counters 128 B / 384 B / 4 KB+128 B / 64 KB+128 B apart (do they share an L2 channel?), 1080p and 4K;
# then the counting kernel's phase clocks (primary rays, counters cleared / carried)
cd "$(dirname "$0")/.." && mkdir -p gpurun_out
V="build_ab/r05_cs32.so build_ab/r05_cs96.so build_ab/r05_cs1056.so build_ab/r05_cs16416.so"
{ echo "== 1080p"; ROUNDS=3 REPS=150 tools/ab2.sh $V
  echo "== 4K"; ROUNDS=3 REPS=60 AB_ARGS="--w 3840 --h 2160" tools/ab2.sh $V; } > gpurun_out/r05_counter_stride_ab.log 2>&1
cat gpurun_out/r05_counter_stride_ab.log
python tools/wave_timeline.py --count --json gpurun_out/r05_count_timeline_cleared.json > /dev/null 2>gpurun_out/r05_count_tl.err &&
python tools/wave_timeline.py --count --carry --json gpurun_out/r05_count_timeline_carried.json > /dev/null 2>>gpurun_out/r05_count_tl.err &&
python - <<'PY'
import json
for f in ("cleared", "carried"):
    s = json.load(open(f"gpurun_out/r05_count_timeline_{f}.json"))
    print(f, {k: s[k] for k in ("kernel_us_plain_build", "kernel_us_timeline_build", "total_rounds", "active_lanes_per_round", "cycles_per_round", "counting", "waves")})
PY
