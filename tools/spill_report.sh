#!/bin/bash
# Kernels of the Sinkhorn translation units that use scratch memory (register spills): usage tools/spill_report.sh [parts...]
cd "$(dirname "$0")/../pilot_amd/csrc"
for part in ${@:-0 1 2 3 4 5 6}; do
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -DSK_PART=$part -Rpass-analysis=kernel-resource-usage -c -o /dev/null sk_inst.hip 2>&1 | python3 -c "
import sys,re
name=None;info={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); info[name]={}
    for key in ('VGPRs','ScratchSize \[bytes/lane\]','Occupancy \[waves/SIMD\]'):
        m=re.search(key+r': (\d+)',l)
        if m and name: info[name][key[:5]]=int(m.group(1))
for n,d in info.items():
    if d.get('Scrat',0)>0: print(n.replace('_ZN5pilot','').replace('EvNS_10GridParamsE',''), d)
"
done
