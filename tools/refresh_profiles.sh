#!/bin/bash
# Run on the GPU box (through gpurun): the evidence profiles/ holds for the current build.
#   bench line; rocprofv3 kernel-trace summary of the same command; PMC passes for the trace kernel; and for the
#   reference's default mode (hit counters live + shadow rays), the scan and the shaded frame: probe logs and a
#   kernel-trace summary each.    usage: tools/refresh_profiles.sh <tag>   -> gpurun_out/<tag>/
set -u
TAG=${1:-final}
OUT=/root/repo/gpurun_out/$TAG
mkdir -p "$OUT"
cd /root/repo
python3 bench.py > "$OUT/bench.json" 2> "$OUT/bench.err" || { echo "bench failed"; tail -5 "$OUT/bench.err"; exit 1; }
python3 tools/show_bench.py "$OUT/bench.json"
python3 tools/scan_probe.py > "$OUT/scan_probe.log" 2>&1; tail -6 "$OUT/scan_probe.log"
python3 tools/count_probe.py > "$OUT/count_probe.log" 2>&1; tail -3 "$OUT/count_probe.log"
python3 tools/shade_probe.py > "$OUT/shade_probe.log" 2>&1; tail -4 "$OUT/shade_probe.log"
python3 tools/default_mode_probe.py > "$OUT/default_mode_probe.log" 2>&1
python3 tools/default_mode_probe.py --carry >> "$OUT/default_mode_probe.log" 2>&1
python3 tools/default_mode_probe.py --fused 1 >> "$OUT/default_mode_probe.log" 2>&1
python3 tools/default_mode_probe.py --fused 0 >> "$OUT/default_mode_probe.log" 2>&1; tail -4 "$OUT/default_mode_probe.log"
python3 tools/wave_timeline.py --json "$OUT/timeline_heavy_1080p.json" > /dev/null 2> "$OUT/timeline.err" || echo "timeline failed"
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace" -- python3 /root/repo/bench.py --steps 50 --warmup 5 --no-cpu-baseline --no-extras > "$OUT/trace.json" 2> "$OUT/trace.err" || echo "kernel trace failed"
find "$OUT/trace" -name "*kernel_stats.csv" -exec cp {} "$OUT/kernel_stats.csv" \;
rocprofv3 --kernel-trace --stats --output-format csv -d "$OUT/trace_default" -- python3 /root/repo/tools/default_mode_probe.py --frames 30 > "$OUT/trace_default.log" 2> "$OUT/trace_default.err" || echo "default-mode kernel trace failed"
find "$OUT/trace_default" -name "*kernel_stats.csv" -exec cp {} "$OUT/default_mode_kernel_stats.csv" \;
cd /root/repo
tools/pmc_profile.sh "$TAG/pmc" > "$OUT/pmc_summary.txt" 2>&1
cp "$OUT/pmc/summary.json" "$OUT/pmc_summary.json" 2>/dev/null
PMC_PROG=tools/default_mode_probe.py PMC_KERNEL=trace_stack_kernel tools/pmc_profile.sh "$TAG/pmc_default" --frames 10 > "$OUT/pmc_default_summary.txt" 2>&1
cp "$OUT/pmc_default/summary.json" "$OUT/pmc_default_summary.json" 2>/dev/null
head -8 "$OUT/kernel_stats.csv"
head -10 "$OUT/default_mode_kernel_stats.csv"
rm -rf "$OUT"/trace "$OUT"/trace_default "$OUT"/pmc/*/ "$OUT"/pmc_default/*/   # keep the summaries (gpurun_out merges at most 64 MiB)
