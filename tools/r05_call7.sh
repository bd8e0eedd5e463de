#!/bin/bash
set -o pipefail
O=gpurun_out/r05c7; mkdir -p $O
echo "== ab 1080p stealing x shares"; ROUNDS=3 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_bal.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_nosteal.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_nosteal.so build_ab/r05_bal.so 2>&1 | tee $O/ab_steal_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=2 REPS=40 timeout -k 10 500 tools/ab2.sh build_ab/r05_bal.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_nosteal.so:SVO_NO_LIST_BALANCE=1 build_ab/r05_nosteal.so 2>&1 | tee $O/ab_steal_4k.log
