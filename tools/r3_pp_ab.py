#!/usr/bin/env python3
"""A/B of the 256 x 256 ping-pong kernel's bias placement (TUTEL_OPT_GEMM_PERSIST: 0 = bias fetched after the K loop, 1 = before it)
on the MFMA-bound shapes, interleaved rounds, bit equality asserted.  Prints one JSON object."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import _lib, ops  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    g = torch.Generator(device=dev).manual_seed(3)
    out = {}
    for name, El, R, K, N, dt in (("8x1024 K=N=2048 bf16", 8, 1024, 2048, 2048, torch.bfloat16), ("4x1024 K=N=2048 bf16 (stage, forced pp)", 4, 1024, 2048, 2048, torch.bfloat16),
                                  ("8x1024 K=N=4096 bf16", 8, 1024, 4096, 4096, torch.bfloat16), ("16x1024 K=N=4096 fp16", 16, 1024, 4096, 4096, torch.float16),
                                  ("64x2048 K=N=2048 bf16", 64, 2048, 2048, 2048, torch.bfloat16)):
        a = torch.randn([El, R, K], device=dev, generator=g).to(dt)
        ws = [(torch.randn([El, N, K], device=dev, generator=g) * 0.03).to(dt) for _ in range(2)]
        b = torch.randn([El, N], device=dev, generator=g).to(dt)
        flops = 2.0 * El * R * K * N
        iters = max(6, int(1.5e-3 / (flops / 1.1e15)))
        ops.set_option(_lib.OPT_GEMM_TILE, 4)
        res, ref = {"late": [], "early": []}, None
        for rnd in range(4):
            for tag, opt in (("late", 0), ("early", 1)):
                ops.set_option(_lib.OPT_GEMM_PERSIST, opt)
                y = ops.expert_gemm(a, ws[0], b, True, act="relu")
                if ref is None:
                    ref = y.clone()
                assert torch.equal(y, ref), (name, tag)
                for i in range(4):
                    ops.expert_gemm(a, ws[i & 1], b, True, act="relu")
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                torch.cuda.synchronize()
                s.record()
                for i in range(iters):
                    ops.expert_gemm(a, ws[i & 1], b, True, act="relu")
                e.record()
                torch.cuda.synchronize()
                res[tag].append(round(s.elapsed_time(e) * 1e3 / iters, 2))
        out[name] = {"us": res, "tflops_best": {t: round(flops / min(v) * 1e-6, 1) for t, v in res.items()}}
    ops.set_option(_lib.OPT_GEMM_TILE, -1)
    ops.set_option(_lib.OPT_GEMM_PERSIST, -1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
