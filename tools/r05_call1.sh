#!/bin/bash
# round 5, GPU call 1: parity of the LUT walk, A/B against the round-4 walk, light and heavy timelines
set -o pipefail
O=gpurun_out/r05c1; mkdir -p $O
echo "== parity" ; timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc $?"; tail -3 $O/pytest.log
echo "== ab 1080p"; ROUNDS=3 REPS=100 timeout -k 10 300 tools/ab2.sh build_ab/r05_base.so build_ab/r05_lut.so 2>&1 | tee $O/ab_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=2 REPS=40 timeout -k 10 300 tools/ab2.sh build_ab/r05_base.so build_ab/r05_lut.so 2>&1 | tee $O/ab_4k.log
echo "== timelines"
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_light_1080p.json > $O/tl_light_1080p.log 2>&1; echo rc $?
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --w 3840 --h 2160 --json $O/tl_light_4k.json > $O/tl_light_4k.log 2>&1; echo rc $?
timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_heavy_lut_1080p.json > $O/tl_heavy_lut_1080p.log 2>&1; echo rc $?
SVO_HIP_LIB=$PWD/build_ab/r05_base.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_heavy_base_1080p.json > $O/tl_heavy_base_1080p.log 2>&1; echo rc $?
python - <<'PY'
import json
for n in ("tl_light_1080p","tl_light_4k","tl_heavy_lut_1080p","tl_heavy_base_1080p"):
    try:
        d=json.load(open(f"gpurun_out/r05c1/{n}.json"))
        print(n, d["kernel_us_plain_build"], d["kernel_us_timeline_build"], "cam", d["camera_walk_us"], "entry", d["loop_entry_us"], "end", d["end_us"], "dry", d["dry_us"])
    except Exception as e: print(n, "failed", e)
PY
