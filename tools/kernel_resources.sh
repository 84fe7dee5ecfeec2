#!/bin/bash
# Register / LDS / spill figures of the engine's kernels as hipcc sees them (CPU box, no GPU needed).
# usage: tools/kernel_resources.sh [name-filter-regex]
cd "$(dirname "$0")/../voxtral_c_amd" || exit 1
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wno-unused-value -fno-slp-vectorize -c csrc/vox_hip_engine.hip -o /tmp/_kr.o \
    -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re
flt=re.compile(sys.argv[1]) if len(sys.argv)>1 else None
cur=None; rows={}
for l in sys.stdin:
    m=re.search(r'remark: .*Function Name: (\S+)',l)
    if m: cur=m.group(1); rows[cur]={}; continue
    m=re.search(r'remark: .*?\s+(SGPRs|VGPRs|AGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|SGPRs Spill|VGPRs Spill|LDS Size \[bytes/block\]): (\d+)',l)
    if m and cur: rows[cur][m.group(1)]=int(m.group(2))
import subprocess
for k,v in rows.items():
    name=subprocess.run(['c++filt',k],capture_output=True,text=True).stdout.strip()
    if flt and not flt.search(name): continue
    print(f\"{name[:110]:110s} v={v.get('VGPRs')} a={v.get('AGPRs')} s={v.get('SGPRs')} scratch={v.get('ScratchSize [bytes/lane]')} occ={v.get('Occupancy [waves/SIMD]')} vspill={v.get('VGPRs Spill')}\")
" "$@"
