#!/bin/bash
set -o pipefail
O=gpurun_out/r05c16; mkdir -p $O
echo "== parity"; timeout -k 10 400 python -m pytest tests/test_parity_gpu.py -m gpu -x -q > $O/pytest.log 2>&1; echo rc $?; tail -2 $O/pytest.log
echo "== ab 1080p (refill 32): stamps by plain stores vs the committed build"; ROUNDS=3 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_cur.so build_ab/r05_new.so 2>&1 | tee $O/ab_new_1080p.log
echo "== ab 1080p block 16x4"; AB_ARGS="--block 4" ROUNDS=3 REPS=120 timeout -k 10 600 tools/ab2.sh build_ab/r05_new.so 2>&1 | tee $O/ab_block4_1080p.log
echo "== ab 4k"; AB_ARGS="--w 3840 --h 2160" ROUNDS=2 REPS=40 timeout -k 10 600 tools/ab2.sh build_ab/r05_cur.so build_ab/r05_new.so 2>&1 | tee $O/ab_new_4k.log
AB_ARGS="--w 3840 --h 2160 --block 4" ROUNDS=2 REPS=40 timeout -k 10 600 tools/ab2.sh build_ab/r05_new.so 2>&1 | tee $O/ab_block4_4k.log
echo "== bench.py"
timeout -k 10 400 python bench.py --steps 20 --warmup 5 > $O/bench.json 2> $O/bench.err; echo rc $?; python tools/show_bench.py $O/bench.json 2>/dev/null | head
