#!/bin/bash
# Developer helper: static VALU / LDS / VMEM instruction counts per kernel of one HIP source (device-only -S), name filter as regex.
# usage: scripts/kernel_valu.sh dasp_pytorch_amd/csrc/reverb.hip 'fb_fused|conv_' [extra -D flags]
src=$1; filt=${2:-.}; shift; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize -Wno-inline-asm -Wno-pass-failed --cuda-device-only -S "$src" -o /tmp/kv.s "$@" 2>/dev/null
python3 - "$filt" <<'PY'
import re, sys, subprocess
from collections import Counter
cur = None; bodies = {}
for l in open('/tmp/kv.s'):
    m = re.match(r'^(_Z\w+):', l)
    if m: cur = m.group(1); bodies[cur] = []; continue
    if cur and l.startswith('.Lfunc_end'): cur = None; continue
    if cur and l.startswith('\t') and not l.strip().startswith((';', '.')): bodies[cur].append(l.split()[0])
names = subprocess.run(['c++filt'], input='\n'.join(bodies), capture_output=True, text=True).stdout.split('\n')
for (k, v), n in zip(bodies.items(), names):
    n = re.sub(r'\(.*', '', n).replace('void dasp::', '')
    if not re.search(sys.argv[1], n): continue
    c = Counter(v)
    valu = sum(x for i, x in c.items() if i.startswith('v_'))
    pk = sum(x for i, x in c.items() if i.startswith('v_pk_'))
    lds = sum(x for i, x in c.items() if i.startswith('ds_'))
    vmem = sum(x for i, x in c.items() if i.startswith(('global_', 'buffer_', 'scratch_', 'flat_')))
    print(f"{n[:56]:56s} VALU {valu:5d} (packed {pk:4d})  LDS {lds:4d}  VMEM {vmem:4d}  total {len(v):5d}")
PY
