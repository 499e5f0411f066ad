"""Dev tool: A/B the GEMM tile choice on pipeline-stage shapes, interleaved and repeated (short
back-to-back timings of one variant drift with the clock state)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops, _lib  # noqa: E402


def main():
    M = H = 2048
    g = torch.Generator().manual_seed(0)
    w1 = (torch.randn([32, H, M], generator=g) / 45).bfloat16().cuda()
    b1 = torch.randn([32, H], generator=g).bfloat16().cuda()
    for El, R in ((16, 256), (8, 512), (4, 1024), (32, 256), (8, 1024)):
        a = torch.randn([El, R, M], generator=g).bfloat16().cuda()
        res = {}
        for rep in range(3):
            for name, opt in (("128", 0), ("256x128", 2), ("256x128 ring3", 3), ("256x256", 1)):
                ops.set_option(_lib.OPT_GEMM_TILE, opt)
                for _ in range(20):
                    ops.expert_gemm(a, w1[:El], b1[:El], True, act="relu")
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(100):
                    ops.expert_gemm(a, w1[:El], b1[:El], True, act="relu")
                e.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(s.elapsed_time(e) * 10)
        ops.set_option(_lib.OPT_GEMM_TILE, -1)
        fl = 2 * El * R * M * H
        print(El, R, {k: [round(v, 1) for v in vs] for k, vs in res.items()},
              {k: round(fl / min(vs) * 1e-6) for k, vs in res.items()}, flush=True)


if __name__ == "__main__":
    main()
