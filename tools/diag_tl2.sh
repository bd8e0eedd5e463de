#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/tl
for s in terrain config2 config3b; do for pt in 0 1; do
  timeout -k 10 300 python tools/wave_timeline.py --scene $s --opt PAIR_TABLE=$pt --opt CAMERA_SHORTCUT=0 --json gpurun_out/tl/${s}_p$pt.json > gpurun_out/tl/${s}_p$pt.log 2>&1 || { echo "timeline $s failed"; tail -5 gpurun_out/tl/${s}_p$pt.log; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/tl/${s}_p$pt.json"))
print("$s pairs $pt", d["kernel_us_plain_build"], d["cycles_per_round"], d["descent"])
PY
done; done
