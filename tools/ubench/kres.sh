#!/bin/bash
# per-kernel register / spill / occupancy table of a .hip file (compiler remarks): tools/ubench/kres.sh file.hip [name filter]
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -c "$1" -o /tmp/kres.o -Rpass-analysis=kernel-resource-usage ${EXTRA} 2>&1 | python3 -c "
import sys,re,subprocess
cur=None; rows=[]
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln) or re.search(r' Name: (\S+)',ln)
    if m: cur={'name':m.group(1)}; rows.append(cur); continue
    m=re.search(r'remark:\s+([A-Za-z ]+?)(?: \[[^\]]*\])?: (\d+)',ln)
    if m and cur is not None: cur[m.group(1).strip()]=int(m.group(2))
flt=sys.argv[1] if len(sys.argv)>1 else ''
for r in rows:
    try: nm=subprocess.run(['/opt/rocm/lib/llvm/bin/llvm-cxxfilt',r['name']],capture_output=True,text=True).stdout.strip().split('(')[0]
    except Exception: nm=r['name']
    if flt and flt not in nm: continue
    print('%-110s VGPR %3s  spill %3s  scratch %4s  occ %s  LDS %s' % (nm[:110], r.get('VGPRs'), r.get('VGPRs Spill'), r.get('ScratchSize'), r.get('Occupancy'), r.get('LDS Size')))
" "$2"
