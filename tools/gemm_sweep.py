"""Dev tool: A/B the expert-GEMM kernel choices at the headline shape (E=64, R=128, M=H=2048) and at
R = 1024 rows per expert.  Variants are switched in-process (tutel_amd_set_option), measured
interleaved and repeated; fc1 (k-major) and fc2 (n-major) weight sets alternate inside the timed loop
so the weights really stream from HBM (a warm Infinity Cache hides layout effects otherwise)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops, _lib  # noqa: E402

VARIANTS = (("128 reg-staged", 0, 0), ("128 LDS-DMA", 1, 0), ("256x128", -1, 2), ("256x256", -1, 1), ("auto", -1, -1))


def main():
    M = H = 2048
    g = torch.Generator().manual_seed(0)
    for E, R in ((64, 128), (8, 1024)):
        a = torch.randn([E, R, M], generator=g).bfloat16().cuda()
        w1 = (torch.randn([E, H, M], generator=g) / 45).bfloat16().cuda()  # k-major [E, N, K]
        w2 = (torch.randn([E, H, M], generator=g) / 45).bfloat16().cuda()  # n-major [E, K, N]
        b = torch.randn([E, H], generator=g).bfloat16().cuda()
        res = {}
        for rep in range(3):
            for name, impl, tile in VARIANTS:
                ops.set_option(_lib.OPT_GEMM_IMPL, impl)
                ops.set_option(_lib.OPT_GEMM_TILE, tile)
                for _ in range(5):
                    h = ops.expert_gemm(a, w1, b, True, act="relu")
                    ops.expert_gemm(h, w2, b, False)
                torch.cuda.synchronize()
                ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
                t1 = t2 = 0.0
                for _ in range(30):
                    ev[0].record()
                    h = ops.expert_gemm(a, w1, b, True, act="relu")
                    ev[1].record()
                    ops.expert_gemm(h, w2, b, False)
                    ev[2].record()
                    torch.cuda.synchronize()
                    t1 += ev[0].elapsed_time(ev[1])
                    t2 += ev[1].elapsed_time(ev[2])
                res.setdefault(name, []).append((round(t1 / 30 * 1e3, 1), round(t2 / 30 * 1e3, 1)))
        ops.set_option(_lib.OPT_GEMM_IMPL, -1)
        ops.set_option(_lib.OPT_GEMM_TILE, -1)
        print(f"E={E} R={R}: (fc1 k-major us, fc2 n-major us) per repetition")
        for name, v in res.items():
            print(f"  {name:16s} {v}", flush=True)


if __name__ == "__main__":
    main()
