#!/bin/bash
# memory-side and L1 counters of the trace kernel for two library builds, one counter group per rocprofv3 pass:
#   tools/pmc_ab.sh <outdir-under-gpurun_out> build_ab/a.so build_ab/b.so
set -u
OUT=/root/repo/gpurun_out/$1; shift
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  tag=$(basename "$lib" .so)
  export SVO_HIP_LIB=/root/repo/$lib
  for grp in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_PENDING_STALL_CYCLES_sum" "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum TCC_EA0_RD_UNCACHED_32B_sum" "TCC_REQ_sum TCC_HIT_sum TCC_MISS_sum TCC_TAG_STALL_sum" "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum" "TCC_EA0_ATOMIC_sum TCC_EA0_ATOMIC_LEVEL_sum"; do
    name=${tag}_$(echo $grp | cut -d' ' -f1)
    timeout -k 10 150 rocprofv3 --pmc $grp --output-format csv -d "$OUT/$name" -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed rc=$?"
  done
done
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json, os
out = sys.argv[1]
res = collections.defaultdict(dict)
for d in sorted(glob.glob(out + "/*/")):
    tag = os.path.basename(d.rstrip("/")).split("_TC")[0]
    agg = collections.defaultdict(lambda: [0.0, 0])
    for f in glob.glob(d + "/*/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            if "trace_stack_kernel" in r["Kernel_Name"]:
                a = agg[r["Counter_Name"]]; a[0] += float(r["Counter_Value"]); a[1] += 1
    for k, v in agg.items():
        res[tag][k] = round(v[0] / v[1], 1)
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"/*/
