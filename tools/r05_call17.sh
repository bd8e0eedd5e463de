#!/bin/bash
set -o pipefail
O=gpurun_out/r05c17; mkdir -p $O
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_light.json --raw $O/tl_light_raw.npz > $O/tl_light.log 2>&1; echo rc $?
timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_heavy.json > $O/tl_heavy.log 2>&1; echo rc $?
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --w 3840 --h 2160 --json $O/tl_light_4k.json > $O/tl_light_4k.log 2>&1; echo rc $?
python - <<'PY'
import json
for n in ("tl_light","tl_heavy","tl_light_4k"):
    d=json.load(open(f"gpurun_out/r05c17/{n}.json"))
    print(n, d["kernel_us_plain_build"], "entry", d["loop_entry_us"], "end", d["end_us"], "dry", d["dry_us"], "rounds", d["rounds_per_wave"], d["total_rounds"], "active", d["active_lanes_per_round"], "refills/round", d["refills_per_round"])
    if n=="tl_heavy": print(d["cycles_per_round"], d["descent"])
    if n=="tl_light": print([(r["t_us"],r["strips_generated"],r["waves_in_loop"]) for r in d["progress_10us"]])
PY
