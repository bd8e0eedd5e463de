#!/bin/bash
set -o pipefail
O=gpurun_out/r05c20; mkdir -p $O
for v in light light_c16; do
SVO_HIP_LIB=$PWD/build_ab/r05_$v.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_$v.json > $O/tl_$v.log 2>&1; echo rc $?
done
python - <<'PY'
import json
for n in ("light","light_c16"):
    d=json.load(open(f"gpurun_out/r05c20/tl_{n}.json"))
    print(n, d["kernel_us_plain_build"], "end", d["end_us"], "answer cycles", d["claim_answer_cycles_pct"], "total", d["claim_total_cycles_pct"])
PY
