#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/tl
for g in 1536 768 256; do
  timeout -k 10 300 python tools/wave_timeline.py --scene terrain --opt GRID_BLOCKS=$g --json gpurun_out/tl/grid$g.json > gpurun_out/tl/grid$g.log 2>&1 || { echo "failed"; tail -5 gpurun_out/tl/grid$g.log; }
  python3 - <<PY
import json
d=json.load(open("gpurun_out/tl/grid$g.json"))
print("grid $g", {k:d[k] for k in ["kernel_us_plain_build","kernel_us_timeline_build","waves","total_rounds","active_lanes_per_round","cycles_per_round","shader_clock_ghz_in_kernel"]}, d["descent"]["cycles_per_wave_iter"], "us/round", d["us_per_round"][4], "rounds/wave", d["rounds_per_wave"][4])
PY
done
