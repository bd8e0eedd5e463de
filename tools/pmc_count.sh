#!/bin/bash
# instruction and atomic counters of the trace kernel per mode (static / shadow rays / counters live, cleared and carried), one counter group per
# rocprofv3 pass:   tools/pmc_count.sh <outdir-under-gpurun_out>
set -u
OUT=/root/repo/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
run() {  # tag, program arguments
  tag=$1; shift
  for grp in "SQ_INSTS_VALU SQ_INSTS_SALU" "SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_VALU" \
             "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum"; do
    name=${tag}__$(echo $grp | cut -d' ' -f1)
    timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 "$@" > "$OUT/$name.log" 2> "$OUT/$name.err" || echo "pass $name failed rc=$?"
  done
}
run shade /root/repo/tools/shade_probe.py
run cleared /root/repo/tools/count_probe.py --reps 8 --only 1
run carried /root/repo/tools/count_probe.py --reps 8 --only 2
run default /root/repo/tools/default_mode_probe.py --frames 12
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, os, re
out = sys.argv[1]
res = collections.defaultdict(lambda: collections.defaultdict(dict))
for d in sorted(glob.glob(out + "/*/")):
    tag = os.path.basename(d.rstrip("/")).split("__")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            m = re.search(r"trace_stack_kernel<([^>]*)>", r["Kernel_Name"])
            if m:
                a = agg[(m.group(1), r["Counter_Name"])]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for (k, c), v in agg.items():
        res[tag][k][c] = round(v[0] / v[1], 1); res[tag][k]["launches"] = v[1]
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"/*/
