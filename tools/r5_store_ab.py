#!/usr/bin/env python3
"""Round 5 A/B: how the grouped GEMMs' output tile leaves the CU (TUTEL_OPT_GEMM_STORE: 0 plain write-back stores, 1 write-through
sc0 sc1, 2 non-temporal) -- the fc1 -> fc2 pair at the headline shape and at the MFMA-bound shapes of an 8-way expert-parallel rank
(one 256-row tile per CU, every block finishing together: the whole output is dirty in L2 when the last wave ends), interleaved,
alternating two weight sets; outputs compared bit for bit between the modes.  Then the whole forward: bench.py once per mode.

    python tools/r5_store_ab.py  -> gpurun_out/r5_store_ab.json"""
import json
import os
import subprocess
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
MODES = (("plain", 0), ("write-through", 1), ("non-temporal", 2))


def bench(fn, iters=40):
    for i in range(8):
        fn(i)
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(iters):
        fn(i)
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters


def main():
    from tutel_amd import _lib, ops
    dev = torch.device("cuda", 0)
    out = {"gemm_pair_us": {}, "forward": {}}
    g = torch.Generator(device=dev).manual_seed(3)
    for name, (E, R, H, M) in (("headline 64 x 128 x 2048 x 2048", (64, 128, 2048, 2048)), ("rank 8 x 1024 x 2048 x 2048", (8, 1024, 2048, 2048)),
                               ("stage 4 x 1024 x 2048 x 2048", (4, 1024, 2048, 2048)), ("configs[3] stage 4 x 1024 x 4096 x 4096", (4, 1024, 4096, 4096))):
        x = torch.randn([E, R, M], device=dev, generator=g).bfloat16()
        w1 = [(torch.randn([E, H, M], device=dev, generator=g) * 0.03).bfloat16() for _ in range(2)]
        w2 = [(torch.randn([E, M, H], device=dev, generator=g) * 0.03).bfloat16() for _ in range(2)]
        b1, b2 = torch.randn([E, H], device=dev, generator=g).bfloat16(), torch.randn([E, M], device=dev, generator=g).bfloat16()

        def pair(i):
            h = ops.expert_gemm(x, w1[i & 1], b1, True, act="relu")     # [E, R, H], weights k-major [E, H, M]
            return ops.expert_gemm(h, w2[i & 1], b2, True)              # [E, R, M], weights k-major [E, M, H]

        res, ref = {}, None
        for rep in range(3):
            for label, mode in MODES:
                ops.set_option(_lib.OPT_GEMM_STORE, mode)
                y = pair(0)
                ref = y.clone() if ref is None else ref
                assert torch.equal(y, ref), f"{name}: store mode {label} changes the output"
                res.setdefault(label, []).append(round(bench(pair), 2))
        ops.set_option(_lib.OPT_GEMM_STORE, -1)
        out["gemm_pair_us"][name] = res
        print(name, res, flush=True)
        del x, w1, w2, b1, b2
    for rep in range(2):
        for label, mode in MODES:
            env = dict(os.environ, TUTEL_AMD_GEMM_STORE=str(mode))
            r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "20", "--warmup", "5", "--no_cpu_baseline", "--no_extra"],
                               env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
            line = [ln for ln in r.stdout.splitlines() if ln.startswith("{")]
            if not line:
                out["forward"].setdefault(label, []).append({"error": r.stderr[-400:]})
                continue
            d = json.loads(line[0])
            rec = {"ms_per_step": d["ms_per_step"], "eager_ms": d["launch_modes"]["other"]["ms_per_step"], "stages_us": d["stages"]["avg_us_per_step"]}
            out["forward"].setdefault(label, []).append(rec)
            print(label, rec, flush=True)
    d = os.path.join(ROOT, "gpurun_out")
    os.makedirs(d, exist_ok=True)
    json.dump(out, open(os.path.join(d, "r5_store_ab.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
