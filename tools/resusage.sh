#!/bin/bash
# developer tool: resource usage of every kernel of one csrc file (the compiler's kernel-resource-usage remarks)
# usage: tools/resusage.sh train_tile.hip [extra hipcc flags]  -> one line per kernel: name | VGPRs AGPRs SGPRs scratch occupancy LDS
cd "$(dirname "$0")/../clid-slam_amd/csrc" || exit 1
f=$1; shift
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -munsafe-fp-atomics -ffp-contract=on -Wno-unused-result -Wno-unused-value -Rpass-analysis=kernel-resource-usage "$@" -c "$f" -o /tmp/ru.$$.o 2>&1 | python3 -c "
import sys, re, subprocess
cur = {}
rows = []
for l in sys.stdin:
    m = re.search(r'remark:\s+(Function Name|VGPRs|AGPRs|TotalSGPRs|ScratchSize \[bytes/lane\]|Occupancy \[waves/SIMD\]|LDS Size \[bytes/block\]):\s*(\S+)', l)
    if not m:
        continue
    k, v = m.group(1), m.group(2)
    if k == 'Function Name':
        if cur: rows.append(cur)
        cur = {'fn': v}
    else:
        cur[k.split(' ')[0]] = v
if cur: rows.append(cur)
for r in rows:
    name = subprocess.run(['c++filt', r['fn']], capture_output=True, text=True).stdout.strip().split('(')[0]
    print(name, '|', ' '.join(f'{k}={v}' for k, v in r.items() if k != 'fn'))
"
rm -f /tmp/ru.$$.o
