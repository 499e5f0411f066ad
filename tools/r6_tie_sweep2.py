"""round 6: second randomized sweep of the tie replay -- the shapes r6_tie_sweep.py leaves out: more than 256 experts (the replay's LDS queue),
small and ragged expert counts, k up to 16, both ATen branches (k * 64 <= E: partial_sort; else nth_element + sort), NaN rows (NaN sorts FIRST in
torch.topk), fp64 scores -- against live CPU torch.topk"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from tutel_amd import ops
g = torch.Generator().manual_seed(606)
res = []
CASES = [  # E, k, dtype, logit scale, rows per iteration, iterations, NaN rows per 1000
    (512, 2, torch.bfloat16, 1.0, 8192, 3, 0), (512, 16, torch.bfloat16, 1.0, 4096, 2, 0), (1024, 4, torch.float16, 2.0, 4096, 2, 0),
    (4096, 2, torch.bfloat16, 1.0, 2048, 1, 0), (8, 2, torch.bfloat16, 0.3, 32768, 3, 0), (48, 3, torch.float16, 0.5, 32768, 3, 0),
    (2, 1, torch.bfloat16, 0.1, 32768, 2, 0), (2, 2, torch.bfloat16, 0.1, 32768, 2, 0), (17, 16, torch.bfloat16, 0.2, 16384, 2, 0),
    (64, 2, torch.bfloat16, 1.0, 32768, 3, 20), (128, 4, torch.float16, 1.0, 32768, 2, 20), (300, 2, torch.bfloat16, 1.0, 8192, 2, 20),
    (64, 2, torch.float64, 1.0, 32768, 2, 0), (96, 6, torch.float32, 1.0, 32768, 2, 5), (640, 10, torch.bfloat16, 1.0, 4096, 2, 0),
]
for E, k, dt, scale, T, iters, nan_per_k in CASES:
    n_rows = n_tied = n_bad = n_nan = 0
    for it in range(iters):
        logits = torch.randn(T, E, generator=g, dtype=torch.float64) * scale
        if dt in (torch.float32, torch.float64):
            logits = (logits * 4).round() / 4          # force ties where the dtype would not produce them
        scores = torch.softmax(logits, dim=1).to(dt)
        if nan_per_k:
            rows = torch.randperm(T, generator=g)[: T * nan_per_k // 1000]
            cols = torch.randint(0, E, (rows.numel(),), generator=g)
            scores[rows, cols] = float("nan")
            two = rows[: rows.numel() // 3]             # a third of them carry a second NaN: NaN == NaN ties among the leaders
            scores[two, torch.randint(0, E, (two.numel(),), generator=g)] = float("nan")
            n_nan += rows.numel()
        want = torch.topk(scores, k, dim=1).indices.int().t()
        got = ops.gate_topk(scores.cuda(), k)[0].cpu()
        top = torch.topk(scores.double().nan_to_num(nan=9.0), min(k + 1, E), dim=1).values
        n_tied += int((top[:, 1:] == top[:, :-1]).any(1).sum())
        n_bad += int((got != want).any(0).sum())
        n_rows += T
    res.append(dict(E=E, k=k, dtype=str(dt), rows=n_rows, rows_with_nan=n_nan, rows_with_ties_among_top_k_plus_1=n_tied, rows_that_differ_from_torch_topk_cpu=n_bad))
    print(res[-1], flush=True)
os.makedirs("gpurun_out", exist_ok=True)
json.dump(dict(source="tools/r6_tie_sweep2.py", torch=torch.__version__, cases=res), open("gpurun_out/r6_tie_sweep2.json", "w"), indent=1)
assert all(r["rows_that_differ_from_torch_topk_cpu"] == 0 for r in res), [r for r in res if r["rows_that_differ_from_torch_topk_cpu"]]
print("ALL EQUAL")
