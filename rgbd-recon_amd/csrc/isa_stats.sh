#!/bin/bash
# developer aid: resource usage + instruction histogram of one kernel (regex on the mangled name)
# usage: isa_stats.sh <file.hip> <mangled-name-regex>
HERE=$(cd "$(dirname "$0")" && pwd)
mkdir -p /tmp/isa && cd /tmp/isa
/opt/rocm/bin/hipcc -O3 --offload-arch=gfx950 -std=c++17 -fPIC -ffp-contract=off -fhip-fp32-correctly-rounded-divide-sqrt \
  -fno-fast-math -Rpass-analysis=kernel-resource-usage -save-temps=obj -c $HERE/$1 -o /tmp/isa/out.o 2>&1 | grep -A10 "$2" | grep -E "VGPRs:|Occupancy|LDS Size|error" | head -4
python3 - "$1" "$2" <<'PY'
import re,sys,glob
from collections import Counter
f=glob.glob('/tmp/isa/%s-hip-amdgcn-amd-amdhsa-gfx950.s' % sys.argv[1].replace('.hip',''))[0]
s=open(f).read()
m=re.search(r'^(\S*%s\S*):[^\n]*\n(.*?)s_endpgm' % sys.argv[2], s, re.S|re.M)
lines=[l.strip() for l in m.group(2).split('\n') if l.strip() and not l.strip().startswith((';','.'))]
c=Counter(l.split()[0] for l in lines)
print(m.group(1), len(lines), 'instructions')
print(', '.join('%s %d'%kv for kv in c.most_common(36)))
PY
