"""print the key numbers of a bench.py JSON line"""
import json, sys
d = json.load(open(sys.argv[1]))
r = d.get("roofline", {})
c = d.get("cpu_baseline", {})
print(d["value"], "Mrays/s", d["ms_per_step"], "ms/step kernel", r.get("kernel_avg_ms"), "ms frac", r.get("frac"),
      "parity", c.get("gpu_frame_matches_oracle_on_sample"), "cpu", c.get("value"))
