"""round 6: what a replayed (tied) row costs the top-k kernel -- T = 4096, E = 64, k = 2, bf16 scores with a chosen number of tied rows per 64-token block"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tutel_amd import _lib, ops
T, E, k = 4096, 64, 2
g = torch.Generator().manual_seed(3)
base = torch.rand(T, E, generator=g).sort(dim=1)[0]            # distinct within a row (fp32)
base = (torch.arange(E).float().unsqueeze(0) * 0.01 + 0.001 + base * 0.005)   # strictly increasing along experts, bf16-distinct
perm = torch.stack([torch.randperm(E, generator=g) for _ in range(T)])
base = base.gather(1, perm).bfloat16()
assert (torch.topk(base.float(), 3, dim=1).values.diff(dim=1) != 0).all()
def with_ties(per_block):
    s = base.clone()
    for b in range(T // 64):
        for r in range(per_block):
            t = b * 64 + (r * 4) % 64 + (r * 4) // 64      # spread over the waves (4 rows per wave)
            top = torch.topk(s[t].float(), 3).indices
            s[t, top[2]] = s[t, top[1]]                     # tie at the k / k+1 boundary
    return s.cuda()
def t_us(x, n=200):
    for _ in range(10): ops.gate_topk(x, k)
    torch.cuda.synchronize(); a = torch.cuda.Event(enable_timing=True); b = torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): ops.gate_topk(x, k)
    b.record(); torch.cuda.synchronize(); return a.elapsed_time(b) / n * 1e3
out = {}
for pb in (0, 1, 2, 4, 8, 15, 16, 32, 64):
    x = with_ties(pb)
    want = torch.topk(x.cpu(), k, dim=1).indices.int().t()
    got = ops.gate_topk(x, k)[0].cpu()
    assert torch.equal(got, want), pb
    ops.set_option(_lib.OPT_TIE_RULE, 0); t0 = t_us(x); ops.set_option(_lib.OPT_TIE_RULE, -1); t1 = t_us(x)
    out[pb] = dict(lowest_index_us=round(t0, 2), torch_cpu_order_us=round(t1, 2))
print(json.dumps(out, indent=1))
