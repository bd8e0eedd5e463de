#!/bin/bash
# vector-memory pipeline counters of the trace kernel (TA / TCP): usage tools/pmc_tcp.sh <outdir-under-gpurun_out>
set -u
OUT=/root/repo/gpurun_out/$1
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  timeout -k 10 120 rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-extras > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed rc=$?"
}
pass t1 TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum
pass t2 TCP_TCP_LATENCY_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum
# (Round 2: a pass with five TA_* counters -- TA_BUSY_avr TA_TOTAL_WAVEFRONTS_sum TA_BUFFER_TOTAL_CYCLES_sum ... -- aborted inside
# rocprofv3 with signal 6.  The cause is on record (gpurun_out/r02_tcp/t3.err of that round): "rocprofiler_create_counter_config ... failed
# with error code 38: Request exceeds the capabilities of the hardware to collect" -- too many TA counters for one pass.  The abort is
# rocprofv3's, before the program starts; the product never ran.  Should TA_* be wanted: ONE counter per pass, under the `timeout` above.)
python3 - "$OUT" <<'PY'
import csv, glob, sys, collections, json
out = sys.argv[1]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if "trace_stack_kernel" in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: round(v[0] / v[1], 1) for k, v in sorted(agg.items())}
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
rm -rf "$OUT"/t?/
