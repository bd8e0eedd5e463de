#!/bin/bash
set -o pipefail
O=gpurun_out/r05c24; mkdir -p $O
echo "== ab 1080p bounds by min3/max3"; ROUNDS=5 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_mm0.so build_ab/r05_mm1.so 2>&1 | tee $O/ab_mm_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=3 REPS=40 timeout -k 10 600 tools/ab2.sh build_ab/r05_mm0.so build_ab/r05_mm1.so 2>&1 | tee $O/ab_mm_4k.log
