#!/bin/bash
set -o pipefail
O=gpurun_out/r05c5; mkdir -p $O
echo "== ab 1080p claim timing x stride (all on column-major ranks)"; ROUNDS=3 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_col.so build_ab/r05_c16.so build_ab/r05_c0.so build_ab/r05_s61.so build_ab/r05_c16s.so build_ab/r05_c0s.so 2>&1 | tee $O/ab_claim_1080p.log
echo "== timeline light col"
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_light_col.json --raw $O/tl_light_col_raw.npz > $O/tl_light_col.log 2>&1; echo rc $?
