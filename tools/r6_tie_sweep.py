"""round 6: randomized sweep of the tie replay -- softmax rows in 16-bit dtypes (natural ties), many shapes, against live CPU torch.topk"""
import sys, os, json, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tutel_amd import ops
g = torch.Generator().manual_seed(2026)
res = []
for E, k, dt, scale in [(64, 2, torch.bfloat16, 1.0), (64, 2, torch.float16, 3.0), (128, 2, torch.float16, 1.0), (128, 2, torch.bfloat16, 0.5), (32, 4, torch.bfloat16, 1.0),
                        (64, 1, torch.bfloat16, 0.3), (16, 3, torch.bfloat16, 0.2), (128, 8, torch.bfloat16, 1.0), (100, 2, torch.bfloat16, 1.0), (256, 2, torch.bfloat16, 1.0),
                        (64, 2, torch.float32, 1.0)]:
    n_rows = n_tied = n_bad = 0
    for it in range(6):
        T = 32768
        logits = (torch.randn(T, E, generator=g) * scale).to(dt)
        if dt == torch.float32:
            logits = (logits * 4).round() / 4          # force ties in fp32
        scores = torch.softmax(logits.float(), dim=1).to(dt)
        want = torch.topk(scores, k, dim=1).indices.int().t()
        got = ops.gate_topk(scores.cuda(), k)[0].cpu()
        top = torch.topk(scores.float(), min(k + 1, E), dim=1).values
        n_tied += int((top[:, 1:] == top[:, :-1]).any(1).sum())
        n_bad += int((got != want).any(0).sum())
        n_rows += T
    res.append(dict(E=E, k=k, dtype=str(dt), rows=n_rows, rows_with_ties_among_top_k_plus_1=n_tied, rows_that_differ_from_torch_topk_cpu=n_bad))
    print(res[-1], flush=True)
assert all(r["rows_that_differ_from_torch_topk_cpu"] == 0 for r in res)
print("ALL EQUAL")
