#!/bin/bash
# register / scratch / occupancy of every trace_stack_kernel instantiation (compiler view), one line each
# template arguments: BLOCK, NS, K, GE, DBG, CNT, SHD, ET
cd /tmp && /opt/rocm/bin/hipcc -O3 -std=c++17 --offload-arch=gfx950 -ffp-contract=off -I/root/repo/include -c --cuda-device-only \
  -Rpass-analysis=kernel-resource-usage /root/repo/octree-tracer_amd/csrc/svo_kernels.hip -o /tmp/kres.o 2>&1 | python3 -c "
import sys,re
cur=None
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: cur={'name':m.group(1)}; continue
    if cur is None: continue
    for k,pat in (('vgpr',r' VGPRs: (\d+)'),('sgpr',r'SGPRs: (\d+)'),('scratch',r'ScratchSize \[bytes/lane\]: (\d+)'),('occ',r'Occupancy \[waves/SIMD\]: (\d+)')):
        m=re.search(pat,l)
        if m: cur[k]=m.group(1)
    if 'LDS Size' in l:
        n=cur['name']
        if '${1:-trace_stack}' in n:
            n=re.sub(r'_ZN3svo18trace_stack_kernelILi256E','stack<',n); n=re.sub(r'EEEvNS.*','>',n); n=n.replace('ELi',',').replace('ELb',',').replace('Li','')
            print(n, 'vgpr',cur.get('vgpr'),'sgpr',cur.get('sgpr'),'scratch',cur.get('scratch'),'occ',cur.get('occ'))
        cur=None
"
