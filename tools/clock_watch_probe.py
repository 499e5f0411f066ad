"""Which clock moves with the operand data?  Runs the headline grouped GEMM in a loop for ~2 s with random and with zero tokens
while sampling `rocm-smi --showclocks --showpower` (read-only).  python tools/clock_watch_probe.py"""
import os
import subprocess
import sys
import threading
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops  # noqa: E402


def sample(stop, out):
    while not stop.is_set():
        r = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True)
        keep = [l.strip() for l in r.stdout.splitlines() if any(k in l for k in ("sclk", "mclk", "fclk", "socclk", "Power"))]
        out.append(" | ".join(k.split(":", 1)[-1].strip() if "GPU[" in k else k for k in keep))
        time.sleep(0.25)


def watch(name, fn, seconds=2.5, per=200):
    stop, out = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, out))
    th.start()
    t0 = time.time()
    n = 0
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    while time.time() - t0 < seconds:
        for _ in range(per):
            fn()
        n += per
        torch.cuda.synchronize()
    e.record()
    torch.cuda.synchronize()
    stop.set()
    th.join()
    print(f"== {name}: {s.elapsed_time(e) * 1e3 / n:.2f} us per launch (incl. sync gaps), {n} launches", flush=True)
    for l in out[2:6]:
        print("   ", l, flush=True)


def big():
    """the MFMA-bound per-rank shape of an 8-way expert-parallel run: the ping-pong kernel vs torch.bmm (hipBLASLt)"""
    E, R, N, K = 8, 1024, 2048, 2048
    g = torch.Generator().manual_seed(0)
    a = torch.randn([E, R, K], generator=g).bfloat16().cuda()
    w = (torch.randn([E, N, K], generator=g) / K ** 0.5).bfloat16().cuda()
    wt = w.transpose(1, 2).contiguous()
    b = torch.zeros([E, N], dtype=torch.bfloat16, device="cuda")
    for name, fn in (("expert_gemm_pp_kernel 8x1024x2048x2048", lambda: ops.expert_gemm(a, w, b, True, act="relu")),
                     ("torch.bmm (hipBLASLt), same shape", lambda: torch.matmul(a, wt)),
                     ("expert_gemm_pp_kernel again", lambda: ops.expert_gemm(a, w, b, True, act="relu"))):
        watch(name, fn)


def impls():
    """the headline shape on the two 128-tile kernels (register-staged vs LDS-DMA), sustained"""
    from tutel_amd import _lib
    E, R, N, K = 64, 128, 2048, 2048
    g = torch.Generator().manual_seed(0)
    a = torch.randn([E, R, K], generator=g).bfloat16().cuda()
    w = (torch.randn([E, N, K], generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.zeros([E, N], dtype=torch.bfloat16, device="cuda")
    wt = w.transpose(1, 2).contiguous()
    for name, impl in (("LDS-DMA kernel", 1), ("register-staged kernel", 0), ("LDS-DMA kernel again", 1)):
        ops.set_option(_lib.OPT_GEMM_IMPL, impl)
        watch(name, lambda: ops.expert_gemm(a, w, b, True, act="relu"))
    ops.set_option(_lib.OPT_GEMM_IMPL, -1)
    watch("torch.bmm (hipBLASLt, no bias / relu)", lambda: torch.matmul(a, wt))


def tiles():
    """8 x 1024 x 2048 x 2048 on each 256-row kernel, sustained"""
    from tutel_amd import _lib
    E, R, N, K = 8, 1024, 2048, 2048
    g = torch.Generator().manual_seed(0)
    a = torch.randn([E, R, K], generator=g).bfloat16().cuda()
    w = (torch.randn([E, N, K], generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.zeros([E, N], dtype=torch.bfloat16, device="cuda")
    for name, t in (("256x256 ping-pong", 4), ("256x256 round-1 kernel", 1), ("256x128 ring of 3", 3), ("128x128 LDS-DMA", 0), ("256x256 ping-pong again", 4)):
        ops.set_option(_lib.OPT_GEMM_TILE, t)
        watch(name, lambda: ops.expert_gemm(a, w, b, True, act="relu"), seconds=2.0)
    ops.set_option(_lib.OPT_GEMM_TILE, -1)


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "tiles":
        return tiles()
    if len(sys.argv) > 1 and sys.argv[1] == "big":
        return big()
    if len(sys.argv) > 1 and sys.argv[1] == "impls":
        return impls()
    E, R, N, K = 64, 128, 2048, 2048
    g = torch.Generator().manual_seed(0)
    w = (torch.randn([E, N, K], generator=g) / K ** 0.5).bfloat16().cuda()
    b = torch.zeros([E, N], dtype=torch.bfloat16, device="cuda")
    for name, t in (("random tokens", torch.randn([E, R, K], generator=g).bfloat16().cuda()),
                    ("zero tokens", torch.zeros([E, R, K], dtype=torch.bfloat16, device="cuda")),
                    ("random tokens", torch.randn([E, R, K], generator=g).bfloat16().cuda())):
        stop, out = threading.Event(), []
        th = threading.Thread(target=sample, args=(stop, out))
        th.start()
        t0 = time.time()
        n = 0
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        while time.time() - t0 < 2.5:
            for _ in range(200):
                ops.expert_gemm(t, w, b, True, act="relu")
            n += 200
            torch.cuda.synchronize()
        e.record()
        torch.cuda.synchronize()
        stop.set()
        th.join()
        print(f"== {name}: {s.elapsed_time(e) * 1e3 / n:.2f} us per launch (incl. sync gaps), {n} launches", flush=True)
        for l in out[2:8]:
            print("   ", l, flush=True)


if __name__ == "__main__":
    main()
