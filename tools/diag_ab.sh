#!/bin/bash
# usage: tools/diag_ab.sh a.so b.so ...   (A/B on terrain, config3b, monu9; then the GPU suite on the in-tree build)
cd /root/repo
for sc in terrain config3b monu9; do
  echo "== $sc"
  if [ $sc = terrain ]; then AB_ARGS="" bash tools/ab.sh "$@"; else AB_ARGS="--scene $sc" bash tools/ab.sh "$@"; fi
done
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
