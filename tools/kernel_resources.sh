#!/bin/bash
# Per-kernel VGPR / scratch / occupancy / LDS of one csrc/*.hip file, as hipcc reports them (no GPU needed).
# usage: tools/kernel_resources.sh conv_mfma.hip
cd "$(dirname "$0")/../pti_ldm_vae_amd/csrc"
hipcc -O3 --offload-arch=gfx950 -fPIC -std=c++17 -ffast-math -fno-finite-math-only -Wno-unused-value -Wno-pass-failed \
  -x hip -c "$1" -o /tmp/kres_$$.o -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None;rows=[]
for l in sys.stdin:
    m=re.search(r'Function Name: (.*)',l)
    if m: cur={'name':m.group(1).strip()}; rows.append(cur)
    for k in ['VGPRs:','AGPRs','ScratchSize','Occupancy','LDS Size','SGPRs:']:
        m=re.search(k+r'[^0-9]*([0-9]+)',l)
        if m and cur is not None: cur[k]=int(m.group(1))
names=subprocess.run(['c++filt'],input='\n'.join(r['name'] for r in rows),capture_output=True,text=True).stdout.split('\n')
for r,n in zip(rows,names):
    n=re.sub(r'\(anonymous namespace\)::','',n)
    n=re.sub(r'\(.*','',n)[:64]
    print(f\"{n:64s} vgpr={r.get('VGPRs:')} agpr={r.get('AGPRs')} scratch={r.get('ScratchSize')} occ={r.get('Occupancy')} lds={r.get('LDS Size')}\")
"
rm -f /tmp/kres_$$.o
