#!/bin/bash
cd /root/repo
for lib in "$@"; do echo "== $lib"; SVO_HIP_LIB=$PWD/$lib python tools/count_probe.py --reps 10 2>&1 | grep -v amdgpu.ids; SVO_HIP_LIB=$PWD/$lib python tools/default_mode_probe.py --frames 12 2>&1 | grep -v amdgpu.ids | cut -c1-200; done
