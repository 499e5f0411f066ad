"""Does the headline grouped GEMM's duration depend on the DATA?  64 experts x 128 rows x 2048 x 2048 bf16, relu, k-major
weights: random / zero tokens x random / zero / constant weights; each 3 x 100 launches, HIP events around the batch."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops  # noqa: E402


def timeit(fn, n=100):
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(n):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / n


def main():
    E, R, N, K = 64, 128, 2048, 2048
    g = torch.Generator().manual_seed(0)
    toks = {"random": torch.randn([E, R, K], generator=g).bfloat16().cuda(), "zero": torch.zeros([E, R, K], dtype=torch.bfloat16, device="cuda")}
    ws = {"random": (torch.randn([E, N, K], generator=g) / K ** 0.5).bfloat16().cuda(),
          "zero": torch.zeros([E, N, K], dtype=torch.bfloat16, device="cuda"),
          "constant 0.01": torch.full([E, N, K], 0.01, dtype=torch.bfloat16, device="cuda"),
          "random sign, |w| = 0.01": (torch.randint(0, 2, [E, N, K], generator=g).float() * 0.02 - 0.01).bfloat16().cuda()}
    b = torch.zeros([E, N], dtype=torch.bfloat16, device="cuda")
    for _ in range(50):
        ops.expert_gemm(toks["random"], ws["random"], b, True, act="relu")
    for tn, t in toks.items():
        for wn, w in ws.items():
            r = sorted(timeit(lambda: ops.expert_gemm(t, w, b, True, act="relu")) for _ in range(3))
            print(f"tokens {tn:7s} weights {wn:24s}: {r[1]:7.2f} us  ({(E * N * K + 2 * E * R * K) * 2 / r[1] * 1e-6:5.2f} TB/s)", flush=True)


if __name__ == "__main__":
    main()
