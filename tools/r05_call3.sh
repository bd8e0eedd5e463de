#!/bin/bash
set -o pipefail
O=gpurun_out/r05c3; mkdir -p $O
echo "== parity (spec)"; SVO_HIP_LIB=$PWD/build_ab/r05_spec.so timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_spec.log 2>&1; echo rc $?; tail -2 $O/pytest_spec.log
echo "== ab 1080p walk"; ROUNDS=4 REPS=120 timeout -k 10 400 tools/ab2.sh build_ab/r05_base.so build_ab/r05_lut2.so build_ab/r05_spec.so build_ab/r05_spec1.so 2>&1 | tee $O/ab_walk_1080p.log
echo "== ab 1080p order"; ROUNDS=4 REPS=120 timeout -k 10 400 tools/ab2.sh build_ab/r05_base.so build_ab/r05_run4.so build_ab/r05_run16.so build_ab/r05_run64.so 2>&1 | tee $O/ab_order_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=3 REPS=40 timeout -k 10 500 tools/ab2.sh build_ab/r05_base.so build_ab/r05_spec.so build_ab/r05_run16.so 2>&1 | tee $O/ab_4k.log
