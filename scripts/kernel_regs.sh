#!/bin/bash
# Developer helper: register / spill / LDS / occupancy figures of the kernels of one HIP source (device-only compile with
# -Rpass-analysis=kernel-resource-usage). usage: scripts/kernel_regs.sh dasp_pytorch_amd/csrc/sosfilt.hip [name filter] [extra -D flags]
src=$1; filt=${2:-.}; shift; shift
hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize --cuda-device-only -c "$src" -o /dev/null -Rpass-analysis=kernel-resource-usage "$@" 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
def flush():
    if cur and re.search(r'$filt', cur.get('name', '')):
        rows.append(dict(cur))
for line in sys.stdin:
    m = re.search(r'remark:\s+([^:]+):\s+(\S+)', line)
    if not m:
        continue
    if m.group(1) == 'Function Name':
        flush(); cur = {'name': m.group(2)}
    else:
        cur[m.group(1).strip()] = m.group(2)
flush()
names = subprocess.run(['c++filt'], input='\n'.join(r['name'] for r in rows), capture_output=True, text=True).stdout.split('\n')
for r, n in zip(rows, names):
    n = re.sub(r'\(.*', '', n).replace('void dasp::', '')
    print(f\"{n[:60]:60s} vgpr {r.get('VGPRs','?'):>4s} agpr {r.get('AGPRs','?'):>3s} sgpr {r.get('TotalSGPRs','?'):>4s} scratch {r.get('ScratchSize [bytes/lane]','?'):>4s} occ {r.get('Occupancy [waves/SIMD]','?'):>2s} lds {r.get('LDS Size [bytes/block]','?'):>6s} sspill {r.get('SGPRs Spill','?'):>3s} vspill {r.get('VGPRs Spill','?')}\")
"
