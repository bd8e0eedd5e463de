#!/bin/bash
# registers / scratch of the default trace_stack_kernel instantiation, compiler view (extra flags pass through)
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -fno-slp-vectorize "$@" -I/root/repo/include -c --cuda-device-only -Rpass-analysis=kernel-resource-usage /root/repo/octree-tracer_amd/csrc/svo_kernels.hip -o /tmp/kres.o > /tmp/kres.txt 2>&1
grep -A14 "Name: _ZN3svo18trace_stack_kernelILi256ELi12ELi3ELb0ELb0ELb0ELb0E" /tmp/kres.txt | grep -i " VGPRs\|Scratch\|SGPRs:" | sed 's/.*remark: *//'
