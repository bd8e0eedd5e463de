#!/bin/bash
set -o pipefail
O=gpurun_out/r05c23; mkdir -p $O
echo "== pipeline probe 4K"; timeout -k 10 400 python tools/pipeline_probe.py --w 3840 --h 2160 --world 8,4,2,1 --frames 120 2>/dev/null | tee $O/pipeline_probe_4k.log
echo "== pipeline probe 1080p"; timeout -k 10 300 python tools/pipeline_probe.py --world 8,4,2,1 --frames 200 2>/dev/null | tee $O/pipeline_probe_1080p.log
echo "== boundary tests (bench line fields, multi-gpu line)"; timeout -k 10 900 python -m pytest tests/test_boundary_gpu.py tests/test_lds_oob_gpu.py -m gpu -x -q > $O/pytest_boundary.log 2>&1; echo rc $?; tail -3 $O/pytest_boundary.log
