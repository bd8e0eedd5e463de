#!/bin/bash
set -o pipefail
O=gpurun_out/r05c2; mkdir -p $O
echo "== ab 1080p"; ROUNDS=4 REPS=120 timeout -k 10 400 tools/ab2.sh build_ab/r05_base.so build_ab/r05_lut.so build_ab/r05_lut2.so build_ab/r05_claim.so 2>&1 | tee $O/ab_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=2 REPS=40 timeout -k 10 400 tools/ab2.sh build_ab/r05_base.so build_ab/r05_lut.so build_ab/r05_lut2.so build_ab/r05_claim.so 2>&1 | tee $O/ab_4k.log
echo "== parity quick (lut2)"; SVO_HIP_LIB=$PWD/build_ab/r05_lut2.so timeout -k 10 300 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest_lut2.log 2>&1; echo rc $?; tail -2 $O/pytest_lut2.log
echo "== timeline raw"
SVO_HIP_LIB=$PWD/build_ab/r05_light.so timeout -k 10 200 python tools/wave_timeline.py --json $O/tl_light_1080p.json --raw $O/tl_light_1080p_raw.npz > $O/tl_light_1080p.log 2>&1; echo rc $?
