#!/bin/bash
set -o pipefail
O=gpurun_out/r05c19; mkdir -p $O
for v in light light_c32 light_c16; do
SVO_HIP_LIB=$PWD/build_ab/r05_$v.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_$v.json --raw $O/tl_${v}_raw.npz > $O/tl_$v.log 2>&1; echo rc $?
done
python - <<'PY'
import json
for n in ("light","light_c32","light_c16"):
    d=json.load(open(f"gpurun_out/r05c19/tl_{n}.json"))
    print(n, d["kernel_us_plain_build"], "end", d["end_us"], "dry", d["dry_us"], "rounds", d["total_rounds"], d["rounds_per_wave"])
    print("  gen/10us", [r["strips_generated"] for r in d["progress_10us"]])
PY
