#!/bin/bash
set -o pipefail
O=gpurun_out/r05c4; mkdir -p $O
echo "== parity (sp32)"; SVO_HIP_LIB=$PWD/build_ab/r05_sp32.so timeout -k 10 300 python -m pytest tests/test_parity_gpu.py tests/test_configs_full.py -m gpu -x -q > $O/pytest_sp32.log 2>&1; echo rc $?; tail -2 $O/pytest_sp32.log
echo "== ab 1080p order"; ROUNDS=4 REPS=120 timeout -k 10 500 tools/ab2.sh build_ab/r05_base.so build_ab/r05_col.so build_ab/r05_sp8.so build_ab/r05_sp32.so build_ab/r05_sp64.so 2>&1 | tee $O/ab_order_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=3 REPS=40 timeout -k 10 500 tools/ab2.sh build_ab/r05_base.so build_ab/r05_col.so build_ab/r05_sp32.so 2>&1 | tee $O/ab_4k.log
