#!/bin/bash
# registers / spills / occupancy of every kernel of csrc/engine.hip as the compiler reports them (no GPU needed)
# usage: tools/kinfo.sh [extra hipcc flags]
cd "$(dirname "$0")/../compressed-image_amd"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-strict-aliasing -mllvm -structurizecfg-skip-uniform-regions=true \
  -Rpass-analysis=kernel-resource-usage "$@" -c csrc/engine.hip -o /tmp/cimg_engine_kinfo.o 2>&1 | python3 -c "
import re,sys
cur=None; rows={}
for line in sys.stdin:
    m=re.search(r'remark: (?:\s*)([A-Za-z ]+?)(?: \[[^\]]*\])?: (.+?) \[-Rpass', line)
    if not m: continue
    k,v=m.group(1).strip(),m.group(2).strip()
    if k=='Function Name': cur=v; rows[cur]={}
    elif cur: rows[cur][k]=v
print('%-30s %6s %6s %6s %8s %8s %8s %5s' % ('kernel','SGPR','VGPR','AGPR','sgprSpl','vgprSpl','scratch','occ'))
for n,r in rows.items():
    print('%-30s %6s %6s %6s %8s %8s %8s %5s' % (n, r.get('TotalSGPRs','?'), r.get('VGPRs','?'), r.get('AGPRs','?'), r.get('SGPRs Spill','?'), r.get('VGPRs Spill','?'), r.get('ScratchSize','?'), r.get('Occupancy','?')))
"
