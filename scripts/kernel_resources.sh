#!/bin/bash
# Registers / scratch / occupancy of every kernel of one translation unit:  bash scripts/kernel_resources.sh knn_f16.hip [filter]
cd "$(dirname "$0")/../kmcuda_amd/csrc"
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -ffp-contract=off -I../../include -c "$1" -o /tmp/kres.o \
  -Rpass-analysis=kernel-resource-usage 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for l in sys.stdin:
    m=re.search(r'remark:\s+(.*?) \[-Rpass', l)
    if not m: continue
    t=m.group(1).strip()
    if t.startswith('Function Name:'):
        cur={'name':t.split(':',1)[1].strip()}; rows.append(cur)
    elif cur is not None and ':' in t:
        k,v=t.split(':',1); cur[k.strip()]=v.strip()
flt=sys.argv[1] if len(sys.argv)>1 else ''
for r in rows:
    n=subprocess.run(['c++filt',r['name']],capture_output=True,text=True).stdout.strip()
    n=n.split('(')[0].replace('void kmx::','')
    if flt and flt not in n: continue
    print('%-70s V=%s A=%s S=%s scratch=%s occ=%s lds=%s'%(n[:70],r.get('VGPRs'),r.get('AGPRs'),r.get('SGPRs'),r.get('ScratchSize [bytes/lane]'),r.get('Occupancy [waves/SIMD]'),r.get('LDS Size [bytes/block]')))
" "${2:-}"
