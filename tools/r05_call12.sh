#!/bin/bash
set -o pipefail
O=gpurun_out/r05c12; mkdir -p $O
echo "== ab 1080p merged short classes"; ROUNDS=3 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_cur.so build_ab/r05_m48.so build_ab/r05_m28.so build_ab/r05_m412.so 2>&1 | tee $O/ab_merge_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=2 REPS=40 timeout -k 10 600 tools/ab2.sh build_ab/r05_cur.so build_ab/r05_m48.so build_ab/r05_m28.so build_ab/r05_m412.so 2>&1 | tee $O/ab_merge_4k.log
echo "== heavy timeline of the current build"
timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_heavy_cur.json > $O/tl_heavy_cur.log 2>&1; echo rc $?
python -c "
import json; d=json.load(open('$O/tl_heavy_cur.json')); print(d['kernel_us_plain_build'], d['cycles_per_round'], d['descent'], 'refills/round', d['refills_per_round'], 'active', d['active_lanes_per_round'], 'rounds', d['total_rounds'], 'clock', d['shader_clock_ghz_in_kernel'])"
