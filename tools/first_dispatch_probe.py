"""What does a short timed region pay on top of K steady-state steps?  After a synchronize: latency of one trivial kernel,
and wall time of n = 1, 2, 5, 20 forwards of the bench layer (busy-polled event at the end), with different idle gaps
before the first launch.  python tools/first_dispatch_probe.py"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def region(fn, n, gap):
    torch.cuda.synchronize()
    if gap:
        time.sleep(gap)
    e = torch.cuda.Event()
    t0 = time.perf_counter()
    for _ in range(n):
        fn()
    t_enq = time.perf_counter()
    e.record()
    while not e.query():
        pass
    t1 = time.perf_counter()
    return (t1 - t0) * 1e6, (t_enq - t0) * 1e6


def main():
    dev = torch.device("cuda", 0)
    layer = bench.build_layer(2048, 2048, 64, 2, 0, 1, torch.bfloat16, False).to(dev).eval()
    torch.manual_seed(0)
    x = torch.randn([16, 256, 2048]).bfloat16().to(dev)
    one = torch.zeros([64], device=dev)
    with torch.no_grad():
        for _ in range(220):
            layer(x)
        for gap in (0, 0.0002, 0.002, 0.05):
            small = sorted(region(lambda: one.add_(1), 1, gap)[0] for _ in range(9))
            print(f"idle gap {gap * 1e3:.1f} ms: one trivial kernel, launch -> observed complete: median {small[4]:.1f} us (min {small[0]:.1f})", flush=True)
            for n in (1, 2, 5, 20, 50):
                r = sorted(region(lambda: layer(x), n, gap) for _ in range(7))
                w, q = r[3]
                print(f"   {n:3d} forwards: wall {w:8.1f} us = {w / n:7.1f} per step, host enqueue {q:7.1f} us; wall - n * 258 = {w - n * 258:6.1f} us", flush=True)
            # the same 20 steps, but the region is entered with the queue already busy (no synchronize before it)
            for _ in range(10):
                layer(x)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                layer(x)
            e1.record()
            torch.cuda.synchronize()
            print(f"   20 forwards entered with a busy queue: {e0.elapsed_time(e1) * 1e3 / 20:.1f} us per step", flush=True)


if __name__ == "__main__":
    main()
