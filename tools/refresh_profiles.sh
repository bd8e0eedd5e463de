#!/bin/bash
# Run on the GPU box (through gpurun): the bench line, the rocprofv3 kernel-trace summary and the PMC passes that
# profiles/ holds for the current build.  usage: tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-final}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
cd /root/repo
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 /root/repo/bench.py --steps 50 --warmup 5 --no-cpu-baseline > "$OUT/trace.json" 2> "$OUT/trace.err" || echo "kernel trace failed"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
cd /root/repo
tools/pmc_profile.sh "$TAG/pmc" > "$OUT/pmc_summary.txt" 2>&1
cp "$OUT/pmc/summary.json" "$OUT/pmc_summary.json" 2>/dev/null
python3 tools/show_bench.py "$OUT/bench.json"
head -8 "$OUT/kernel_stats.csv"
