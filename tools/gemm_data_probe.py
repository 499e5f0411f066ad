"""Dev tool: expert GEMM time vs operand data (random / zero / constant) at a given shape."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops
E, R, M, H = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 1024, 2048, 2048))]
g = torch.Generator().manual_seed(0)
x = torch.randn([E, R, M], generator=g).bfloat16().cuda()
w1 = (torch.randn([E, H, M], generator=g) / 45).bfloat16().cuda()
w2 = (torch.randn([E, H, M], generator=g) / 45).bfloat16().cuda()
b = torch.randn([E, H], generator=g).bfloat16().cuda()
flops = 2 * E * R * M * H
for name, xx, ww1, ww2 in [("random x, random W", x, w1, w2), ("zero x, random W", torch.zeros_like(x), w1, w2),
                           ("random x, const W", x, torch.full_like(w1, 0.0078125), torch.full_like(w2, 0.0078125)),
                           ("zero x, zero W", torch.zeros_like(x), torch.zeros_like(w1), torch.zeros_like(w2)),
                           ("random x, random W", x, w1, w2)]:
    ev = []
    for i in range(40):
        s1, e1, s2, e2 = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        s1.record(); hh = ops.expert_gemm(xx, ww1, b, True, act="relu"); e1.record()
        s2.record(); yy = ops.expert_gemm(hh, ww2, b, False); e2.record()
        if i >= 10: ev.append((s1, e1, s2, e2))
    torch.cuda.synchronize()
    f1 = sum(a.elapsed_time(b_) for a, b_, _, _ in ev) / len(ev) * 1e3
    f2 = sum(c.elapsed_time(d) for _, _, c, d in ev) / len(ev) * 1e3
    print("%-22s fc1 %7.1f us (%6.0f TF)  fc2 %7.1f us (%6.0f TF)" % (name, f1, flops / f1 * 1e-6, f2, flops / f2 * 1e-6))
xt = torch.randn([E, R, M], generator=g).bfloat16().cuda()
wt = (torch.randn([E, M, H], generator=g) / 45).bfloat16().cuda()
for name, a_, b_ in [("torch.bmm random", xt, wt), ("torch.bmm zeros", torch.zeros_like(xt), torch.zeros_like(wt))]:
    for _ in range(5): torch.matmul(a_, b_)
    torch.cuda.synchronize(); s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(30): torch.matmul(a_, b_)
    e.record(); torch.cuda.synchronize(); t = s.elapsed_time(e) * 1e3 / 30
    print("%-22s %7.1f us (%6.0f TF)" % (name, t, flops / t * 1e-6))
