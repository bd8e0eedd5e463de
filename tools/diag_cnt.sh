#!/bin/bash
cd /root/repo
for lib in "$@"; do echo "== $lib"; SVO_HIP_LIB=$PWD/$lib python tools/count_probe.py --reps 6 2>&1 | grep -v amdgpu.ids; done
