#!/bin/bash
cd /root/repo
python tools/repro_sched.py 2>&1 | grep -v amdgpu.ids | grep -vc " ok$"
python tools/repro_sched.py 2>&1 | grep -v amdgpu.ids | grep -v " ok$" | cut -c1-300 | head -5
timeout -k 10 600 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
