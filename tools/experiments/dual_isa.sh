#!/bin/bash
# ISA of one trace_dual_kernel instantiation (default: GE = 0, DBG = 0) -> /tmp/dual_isa.s, and where its loads and waits are
# usage: tools/dual_isa.sh [Lb0ELb0]
sel=${1:-Lb0ELb0}
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I/root/repo/include -S --cuda-device-only \
  /root/repo/octree-tracer_amd/csrc/svo_dual.hip -o /tmp/svo_dual.s 2>&1 | grep -v "warning\|^$" 
python3 - "$sel" <<'PY'
import sys
sel=sys.argv[1]
L=open('/tmp/svo_dual.s').read().split('\n')
start=[i for i,l in enumerate(L) if l.startswith('_ZN3svo17trace_dual_kernel') and sel+'EEEv' in l.split(':')[0] and ': ' in l][0]
end=next(i for i in range(start,len(L)) if 's_endpgm' in L[i])
body=L[start:end+1]
open('/tmp/dual_isa.s','w').write('\n'.join(body))
n=sum(1 for l in body if l.startswith('\t') and not l.startswith('\t.') and not l.startswith('\t;'))
print('instructions:', n)
for i,l in enumerate(body):
    if 'buffer_load_dword ' in l or 's_waitcnt vmcnt' in l or 'scratch_' in l: print(i,l)
PY
