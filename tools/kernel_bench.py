"""Per-kernel timing of the HIP ops at the BASELINE config[1] shape (dev tool, not bench.py).
Prints one JSON line per op: avg microseconds over `iters` launches (HIP events on the current
stream), algorithmic bytes / flops and the implied GB/s or TFLOP/s."""
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops  # noqa: E402


def timeit(fn, iters=50, warmup=10):
    for _ in range(warmup):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters  # us


def main():
    T, M, H, E, k = 4096, 2048, 2048, 64, 2
    dtype = torch.bfloat16
    dev = "cuda"
    g = torch.Generator().manual_seed(0)
    x = torch.randn([T, M], generator=g).to(dtype).to(dev)
    logits = torch.randn([T, E], generator=g).to(dev)
    w1 = (torch.randn([E, H, M], generator=g) / 45).to(dtype).to(dev)
    w2 = (torch.randn([E, H, M], generator=g) / 45).to(dtype).to(dev)
    b1 = torch.randn([E, H], generator=g).to(dtype).to(dev)
    b2 = torch.randn([E, M], generator=g).to(dtype).to(dev)
    C = k * ((T + E - 1) // E)
    s = 2

    idx, gates, ws, scores = ops.gate_topk(logits, k, apply_softmax=True)
    loc, cnt, stats, l_aux, smap = ops.compute_location(idx, E, ws=ws, capacity=C, want_l_aux=True)
    enc = ops.fast_encode(x, smap, None, E * C)
    h = ops.expert_gemm(enc.view(E, C, M), w1, b1, True, act="relu")
    y = ops.expert_gemm(h, w2, b2, False)
    dec = ops.fast_decode(y.view(E * C, M), idx, loc, gates, C)
    n_kept = int((loc < C).sum())

    rows = []

    def rec(name, us, bytes_=None, flops=None):
        r = {"op": name, "us": round(us, 2)}
        if bytes_:
            r["GB/s"] = round(bytes_ / us * 1e-3, 1)
            r["MB"] = round(bytes_ / 1e6, 2)
        if flops:
            r["TFLOP/s"] = round(flops / us * 1e-6, 1)
        rows.append(r)
        print(json.dumps(r), flush=True)

    rec("gate_topk(softmax)", timeit(lambda: ops.gate_topk(logits, k, apply_softmax=True, ws=ws)), T * E * 4)
    rec("compute_location", timeit(lambda: ops.compute_location(idx, E, ws=ws, capacity=C, want_l_aux=True)))
    rec("fast_encode", timeit(lambda: ops.fast_encode(x, smap, None, E * C)), (T + E * C) * M * s)
    rec("fast_decode", timeit(lambda: ops.fast_decode(y.view(E * C, M), idx, loc, gates, C)), (n_kept + T) * M * s)
    gemm_bytes = (E * H * M + E * C * M + E * C * H) * s
    gemm_flops = 2 * E * C * M * H
    rec("expert_gemm1(kmajor,relu,bias)", timeit(lambda: ops.expert_gemm(enc.view(E, C, M), w1, b1, True, act="relu")), gemm_bytes, gemm_flops)
    rec("expert_gemm2(nmajor,bias)", timeit(lambda: ops.expert_gemm(h, w2, b2, False)), gemm_bytes, gemm_flops)
    ev = enc.view(E, C, M)
    w1t = w1.permute(0, 2, 1)
    rec("torch.bmm fc1 (rocBLAS/hipBLASLt yardstick)", timeit(lambda: torch.matmul(ev, w1t)), gemm_bytes, gemm_flops)
    rec("torch.bmm fc2 (yardstick)", timeit(lambda: torch.matmul(h, w2)), gemm_bytes, gemm_flops)
    big = torch.empty([256 * 1024 * 1024], dtype=torch.uint8, device=dev)
    big2 = torch.empty_like(big)
    rec("torch copy 256MiB (HBM yardstick)", timeit(lambda: big2.copy_(big), iters=20), 2 * big.numel())
    # C4-like per-rank shape: E_loc=8, R=1024, M=H=4096 (MFMA-bound regime)
    El, R, M4 = 8, 1024, 4096
    a4 = torch.randn([El, R, M4], generator=g).to(dtype).to(dev)
    w4 = (torch.randn([El, M4, M4], generator=g) / 64).to(dtype).to(dev)
    f4 = 2 * El * R * M4 * M4
    by4 = (El * M4 * M4 + 2 * El * R * M4) * s
    rec("expert_gemm1 C4 shape (8x1024x4096x4096)", timeit(lambda: ops.expert_gemm(a4, w4, None, True, act="relu"), iters=20), by4, f4)
    rec("expert_gemm2 C4 shape", timeit(lambda: ops.expert_gemm(a4, w4, None, False), iters=20), by4, f4)
    rec("torch.bmm C4 shape (yardstick)", timeit(lambda: torch.matmul(a4, w4), iters=20), by4, f4)
    # per-rank shapes of the expert-parallel runs at the headline config (M = H = 2048): N = 2 / 4 / 8 ranks
    for El2, R2 in ((32, 256), (16, 512), (8, 1024)):
        a2 = torch.randn([El2, R2, M], generator=g).to(dtype).to(dev)
        f2 = 2 * El2 * R2 * M * H
        by2 = (El2 * H * M + 2 * El2 * R2 * M) * s
        rec(f"expert_gemm1 EP shape ({El2}x{R2}x{H}x{M})", timeit(lambda: ops.expert_gemm(a2, w1[:El2], b1[:El2], True, act="relu"), iters=20), by2, f2)
        rec(f"expert_gemm2 EP shape ({El2}x{R2})", timeit(lambda: ops.expert_gemm(a2, w2[:El2], b2[:El2], False), iters=20), by2, f2)
        rec(f"torch.bmm EP shape ({El2}x{R2}) (yardstick)", timeit(lambda: torch.matmul(a2, w2[:El2]), iters=20), by2, f2)
    # SwiGLU expert (experts/llama_ffn.py): three launches, silu and the gating product fused into the first two
    from tutel_amd.experts.llama_ffn import LlamaFFNNetwork
    net = LlamaFFNNetwork(M, H, E, 1).to(dtype).to(dev).eval()

    class Ctx:
        group = None
    with torch.no_grad():
        us = timeit(lambda: net(enc.view(E, C, M), Ctx), iters=30)
    rec("llama_ffn SwiGLU expert (3 grouped GEMMs, E=64 x 128 rows)", us, (3 * E * H * M + 2 * E * C * M + 3 * E * C * H) * s, 3 * 2 * E * C * M * H)
    # one pipeline stage of the overlapped all-to-all at degree 2: half the local experts, all W*C rows
    from tutel_amd import _lib
    for El2, R2 in ((16, 256), (8, 512), (4, 1024)):
        a2 = torch.randn([El2, R2, M], generator=g).to(dtype).to(dev)
        f2 = 2 * El2 * R2 * M * H
        by2 = (El2 * H * M + 2 * El2 * R2 * M) * s
        for name, opt in (("auto", -1), ("128-tile", 0), ("256x128", 2), ("256x256", 1)):
            ops.set_option(_lib.OPT_GEMM_TILE, opt)
            rec(f"stage gemm1 ({El2}x{R2}) {name}", timeit(lambda: ops.expert_gemm(a2, w1[:El2], b1[:El2], True, act="relu"), iters=20), by2, f2)
        ops.set_option(_lib.OPT_GEMM_TILE, -1)
    out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out")
    os.makedirs(out, exist_ok=True)
    with open(os.path.join(out, "kernel_bench.json"), "w") as f:
        json.dump(rows, f, indent=1)


if __name__ == "__main__":
    main()
