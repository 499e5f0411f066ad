"""Dev tool: true device time of the routing kernels for several configurations.  N launches are
captured into one HIP graph (no host gaps) and replayed between two events."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops  # noqa: E402


def graph_time(fn, n=40, reps=5):
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s):
            for _ in range(n):
                fn()
        best = 1e9
        for _ in range(reps):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            g.replay()
            b.record()
            torch.cuda.synchronize()
            best = min(best, a.elapsed_time(b) * 1e3 / n)
    return best


def main():
    g = torch.Generator().manual_seed(0)
    for (T, E, k, dt) in [(4096, 64, 2, torch.bfloat16), (4096, 64, 2, torch.float32), (4096, 64, 1, torch.float32),
                          (4096, 16, 2, torch.float32), (1024, 64, 2, torch.float32), (16384, 64, 2, torch.float32),
                          (4096, 256, 2, torch.float32)]:
        logits = torch.randn([T, E], generator=g).to(dt).cuda()
        scores = torch.softmax(logits.float(), 1).to(dt)
        ws = ops.routing_workspace(T, E, k, logits.device)
        C = k * ((T + E - 1) // E)
        smap = torch.empty([E * C], dtype=torch.int32, device="cuda")
        idx, gates, _, _ = ops.gate_topk(logits, k, apply_softmax=True, ws=ws)
        t_sm = graph_time(lambda: ops.gate_topk(logits, k, apply_softmax=True, ws=ws, clear=smap))
        t_ns = graph_time(lambda: ops.gate_topk(scores, k, apply_softmax=False, ws=ws))
        t_loc = graph_time(lambda: ops.compute_location(idx, E, ws=ws, capacity=C, want_l_aux=True, cleared_slot_map=smap))
        t_loc0 = graph_time(lambda: ops.compute_location(idx, E, ws=ws, capacity=0, want_l_aux=False))
        print("T=%5d E=%3d k=%d %-8s  topk(softmax+clear) %6.2f us   topk(scores) %6.2f us   location(+map,+laux) %6.2f us   location(bare) %6.2f us"
              % (T, E, k, str(dt)[6:], t_sm, t_ns, t_loc, t_loc0))
    x = torch.randn([4096, 2048], generator=g).bfloat16().cuda()
    w = torch.randn([64, 2048], generator=g).bfloat16().cuda()
    print("gate linear (hipBLASLt) %.2f us" % graph_time(lambda: torch.nn.functional.linear(x, w)))
    z = torch.zeros([1024], device="cuda")
    print("tiny elementwise kernel %.2f us" % graph_time(lambda: z.add_(1.0)))


if __name__ == "__main__":
    main()
