#!/bin/bash
# PMC passes for the trace kernel (run on the GPU box through gpurun).  Counters are collected in their own runs (--pmc
# only, no tracing domains), one group per pass, as the microarch guide says.
# usage: tools/pmc_profile.sh <outdir-under-gpurun_out> [bench.py arguments...]
#   PMC_PROG=<script under the repo> profiles that program instead of bench.py (its arguments follow the outdir likewise)
#   PMC_KERNEL=<substring> selects the kernel whose counters are summarised (default: trace_stack_kernel)
set -u
OUT=/root/repo/gpurun_out/$1; shift
EXTRA=("$@")   # forwarded to the profiled program (they used to be dropped)
PROG=/root/repo/${PMC_PROG:-bench.py}
KERNEL=${PMC_KERNEL:-trace_stack_kernel}
if [ "${PMC_PROG:-bench.py}" = "bench.py" ]; then BASE=(--steps 10 --warmup 2 --no-cpu-baseline --no-extras); else BASE=(); fi
mkdir -p "$OUT"
cd /tmp && export TMPDIR=/tmp
pass() {
  name=$1; shift
  rocprofv3 --pmc "$@" --output-format csv -d "$OUT/$name" -- python3 "$PROG" "${BASE[@]}" "${EXTRA[@]}" > "$OUT/$name.json" 2> "$OUT/$name.err" || echo "pass $name failed rc=$?"
}
pass fetch FETCH_SIZE
pass write WRITE_SIZE
pass tcc TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum
pass ea TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_32B_sum
pass sq1 SQ_WAVES SQ_INSTS_VALU SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR
pass sq2 SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_THREAD_CYCLES_VALU
pass sq3 SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_INST_CYCLES_VMEM SQ_LDS_BANK_CONFLICT SQ_INSTS_VALU SQ_WAVE_CYCLES
pass grbm GRBM_GUI_ACTIVE
python3 - "$OUT" "$KERNEL" <<'PY'
import csv, glob, sys, collections, json
out, kernel = sys.argv[1], sys.argv[2]
agg = collections.defaultdict(lambda: [0.0, 0])
for f in glob.glob(out + "/*/*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if kernel in r["Kernel_Name"]:
            a = agg[r["Counter_Name"]]
            a[0] += float(r["Counter_Value"]); a[1] += 1
res = {k: {"mean_per_launch": v[0] / v[1], "launches": v[1]} for k, v in sorted(agg.items())}
res["_kernel"] = kernel
res["_command"] = " ".join(sys.argv[3:])
json.dump(res, open(out + "/summary.json", "w"), indent=1)
print(json.dumps(res, indent=1))
PY
