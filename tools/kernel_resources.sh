#!/bin/bash
# Register / scratch use of the kernels of one HIP source whose name matches a pattern:  tools/kernel_resources.sh gemm.hip mfma256w [extra -D flags]
src=$1; pat=$2; shift 2
cd "$(dirname "$0")/../map-dit_amd/csrc"
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -ffp-contract=fast -fno-slp-vectorize -fno-vectorize "$@" \
  -Rpass-analysis=kernel-resource-usage -c $src -o /tmp/kres_$$.o 2>&1 | python3 -c "
import sys,re,subprocess
pat=sys.argv[1]
name=None; rec={}
for ln in sys.stdin:
    m=re.search(r'Function Name: (\S+)',ln)
    if m: name=m.group(1); rec[name]={}
    for k,key in (('VGPRs:','v'),('ScratchSize','scr'),('SGPRs:','s')):
        if k in ln and name and 'Spill' not in ln: rec[name][key]=ln.split(':')[-1].split('[')[0].strip()
for n,r in rec.items():
    if pat in n:
        d=subprocess.run(['c++filt',n],capture_output=True,text=True).stdout.strip()
        d=re.sub(r'\(anonymous namespace\)::','',d); d=re.sub(r'\(.*','',d)
        print(f\"{r.get('v','?'):>4} vgpr {r.get('s','?'):>4} sgpr {r.get('scr','?'):>5} scratch  {d}\")
" "$pat"
rm -f /tmp/kres_$$.o
