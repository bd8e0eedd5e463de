#!/bin/bash
set -o pipefail
O=gpurun_out/r05c11; mkdir -p $O
echo "== full gpu suite (default lib: LUT 2, column-major ranks, list shares, streaming records)"; timeout -k 10 600 python -m pytest tests -m gpu -x -q > $O/pytest.log 2>&1; echo rc $?; tail -2 $O/pytest.log
echo "== ab 1080p streaming record stores"; ROUNDS=4 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_nt0.so build_ab/r05_nt1.so 2>&1 | tee $O/ab_nt_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=3 REPS=40 timeout -k 10 500 tools/ab2.sh build_ab/r05_nt0.so build_ab/r05_nt1.so 2>&1 | tee $O/ab_nt_4k.log
echo "== claim latency probe"
SVO_HIP_LIB=$PWD/build_ab/r05_light_probe.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_probe.json > $O/tl_probe.log 2>&1; echo rc $?
python -c "
import json; d=json.load(open('$O/tl_probe.json')); print('claim wait cycles per wave', d['claim_wait_cycles_per_wave'], 'strips', d['strips_generated'], 'waves', d['waves'], 'clock', d['shader_clock_ghz_in_kernel'], 'end', d['end_us'])"
