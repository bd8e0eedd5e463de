#!/bin/bash
set -o pipefail
O=gpurun_out/r05c10; mkdir -p $O
echo "== ab 1080p shares from mean times"; ROUNDS=3 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_bal4.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_bal4.so 2>&1 | tee $O/ab_bal4_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=2 REPS=40 timeout -k 10 500 tools/ab2.sh build_ab/r05_bal4.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_bal4.so 2>&1 | tee $O/ab_bal4_4k.log
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_light_bal4.json --raw $O/tl_light_bal4_raw.npz > $O/tl_light_bal4.log 2>&1; echo rc $?
