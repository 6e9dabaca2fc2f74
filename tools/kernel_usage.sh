#!/bin/bash
# VGPRs / AGPRs / scratch / occupancy / LDS of every kernel of the given Sinkhorn translation units: tools/kernel_usage.sh [parts...]
cd "$(dirname "$0")/../pilot_amd/csrc"
for part in ${@:-6 7 8 9}; do
  extra=""; case $part in 0|1|2|3|4|5) extra="-mllvm -amdgpu-mfma-vgpr-form=1";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 $extra -DSK_PART=$part -Rpass-analysis=kernel-resource-usage -c -o /dev/null sk_inst.hip 2>&1 | python3 -c "
import sys,re,subprocess
name=None;info={}
for l in sys.stdin:
    m=re.search(r'Function Name: (\S+)',l)
    if m: name=m.group(1); info[name]={}
    for key,short in (('VGPRs','vgpr'),('AGPRs','agpr'),('ScratchSize \[bytes/lane\]','scratch'),('Occupancy \[waves/SIMD\]','occ')):
        m=re.search(r' '+key+r': (\d+)',l)
        if m and name: info[name][short]=int(m.group(1))
for n,d in info.items():
    dn=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
    dn=dn.replace('pilot::','').replace('(GridParams)','').replace('void ','')
    print('part $part  %-70s %s' % (dn, d))
"
done
