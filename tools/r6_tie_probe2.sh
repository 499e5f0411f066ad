#!/bin/bash
# kernel time of gate_topk_quad_kernel by number of tied rows per 64-token block (rocprofv3), round 6
cd /tmp && export TMPDIR=/tmp
for pb in 0 1 3 8 15; do
  out=/tmp/tp_$pb; rm -rf $out; mkdir -p $out
  PB=$pb rocprofv3 --kernel-trace --stats --output-format csv -d $out -o r -- python - <<'PY' > /dev/null 2>&1
import sys, os
sys.path.insert(0, os.environ["GRAFT_REPO_ROOT"])
import torch
from tutel_amd import ops
T, E, k = 4096, 64, 2
g = torch.Generator().manual_seed(3)
base = torch.rand(T, E, generator=g)
base = (torch.arange(E).float().unsqueeze(0) * 0.01 + 0.001 + base * 0.005)
perm = torch.stack([torch.randperm(E, generator=g) for _ in range(T)])
s = base.gather(1, perm).bfloat16()
pb = int(os.environ["PB"])
for b in range(T // 64):
    for r in range(pb):
        t = b * 64 + (r * 4) % 64 + (r * 4) // 64
        top = torch.topk(s[t].float(), 3).indices
        s[t, top[2]] = s[t, top[1]]
x = s.cuda()
for _ in range(300): ops.gate_topk(x, k)
torch.cuda.synchronize()
PY
  f=$(find $out -name "*kernel_stats.csv" | head -1)
  python3 -c "
import csv,sys
for r in csv.DictReader(open('$f')):
    if 'gate_topk_quad' in r['Name']: print('tied rows per block: $pb  calls', r['Calls'], ' avg_us', float(r['AverageNs'])/1e3, ' min_us', float(r['MinNs'])/1e3)
"
done
