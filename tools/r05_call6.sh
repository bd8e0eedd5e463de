#!/bin/bash
set -o pipefail
O=gpurun_out/r05c6; mkdir -p $O
echo "== parity (balance on)"; SVO_HIP_LIB=$PWD/build_ab/r05_bal.so timeout -k 10 400 python -m pytest tests/test_parity_gpu.py tests/test_configs_full.py -m gpu -x -q > $O/pytest_bal.log 2>&1; echo rc $?; tail -2 $O/pytest_bal.log
echo "== ab 1080p list shares"; ROUNDS=4 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_bal.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_bal.so build_ab/r05_ballut.so 2>&1 | tee $O/ab_bal_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=3 REPS=40 timeout -k 10 500 tools/ab2.sh build_ab/r05_bal.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_bal.so build_ab/r05_ballut.so 2>&1 | tee $O/ab_bal_4k.log
echo "== timeline light with shares"
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_light_bal.json --raw $O/tl_light_bal_raw.npz > $O/tl_light_bal.log 2>&1; echo rc $?
