#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/tl
for s in "$@"; do
  timeout -k 10 300 python tools/wave_timeline.py --scene $s --json gpurun_out/tl/$s.json > gpurun_out/tl/$s.log 2>&1 || { echo "timeline $s failed"; tail -5 gpurun_out/tl/$s.log; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/tl/$s.json"))
print("$s", {k:d[k] for k in ["kernel_us_plain_build","kernel_us_timeline_build","total_rounds","active_lanes_per_round","refills_per_round","cycles_per_round","descent"]})
print("   end_us", d["end_us"], "dry_us", d["dry_us"])
PY
done
