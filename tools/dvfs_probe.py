import os, sys, torch, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from tutel_amd import ops
g = torch.Generator().manual_seed(0)
dt = torch.bfloat16
E, C, M, H = 64, 128, 2048, 2048
x = torch.randn([E, C, M], generator=g).to(dt).cuda()
w1 = (torch.randn([E, H, M], generator=g) / 45).to(dt).cuda()
w2 = (torch.randn([E, H, M], generator=g) / 45).to(dt).cuda()
b = torch.randn([E, H], generator=g).to(dt).cuda()
def run(gap_cycles, iters=60):
    ev = []
    for i in range(iters + 10):
        s1, e1, s2, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        s1.record(); h = ops.expert_gemm(x, w1, b, True, act="relu"); e1.record()
        if gap_cycles: torch.cuda._sleep(gap_cycles)
        s2.record(); y = ops.expert_gemm(h, w2, b, False); e2.record()
        if gap_cycles: torch.cuda._sleep(gap_cycles)
        if i >= 10: ev.append((s1, e1, s2, e2))
    torch.cuda.synchronize()
    f1 = sum(a.elapsed_time(b_) for a, b_, _, _ in ev) / len(ev) * 1e3
    f2 = sum(c.elapsed_time(d) for _, _, c, d in ev) / len(ev) * 1e3
    return f1, f2
for gap in (0, 100000, 400000, 2000000, 0):
    f1, f2 = run(gap)
    print("gap cycles %8d: fc1 %.1f us  fc2 %.1f us" % (gap, f1, f2))
# fc2 alone back-to-back, fc1 alone back-to-back
h = ops.expert_gemm(x, w1, b, True, act="relu")
def alone(fn, iters=60):
    for _ in range(10): fn()
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize(); return s.elapsed_time(e) * 1e3 / iters
print("fc1 alone %.1f  fc2 alone %.1f" % (alone(lambda: ops.expert_gemm(x, w1, b, True, act="relu")), alone(lambda: ops.expert_gemm(h, w2, b, False))))

# ---- data dependence (power/clock): same kernels on constant operands
for name, xx, ww1, ww2 in [("random x, random W", x, w1, w2),
                           ("zero x, random W", torch.zeros_like(x), w1, w2),
                           ("random x, constant W", x, torch.full_like(w1, 0.0078125), torch.full_like(w2, 0.0078125)),
                           ("zero x, zero W", torch.zeros_like(x), torch.zeros_like(w1), torch.zeros_like(w2))]:
    ev = []
    for i in range(50):
        s1, e1, s2, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        s1.record(); hh = ops.expert_gemm(xx, ww1, b, True, act="relu"); e1.record()
        s2.record(); yy = ops.expert_gemm(hh, ww2, b, False); e2.record()
        if i >= 10: ev.append((s1, e1, s2, e2))
    torch.cuda.synchronize()
    print("%-24s fc1 %.1f us  fc2 %.1f us" % (name, sum(a.elapsed_time(b_) for a, b_, _, _ in ev) / len(ev) * 1e3,
                                              sum(c.elapsed_time(d) for _, _, c, d in ev) / len(ev) * 1e3))
