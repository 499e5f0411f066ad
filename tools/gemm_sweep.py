"""Dev tool: A/B the expert-GEMM tile configurations (env TUTEL_AMD_GEMM_CFG) in separate
processes, interleaved rounds, at the headline shape and the C4-like shape."""
import json
import os
import subprocess
import sys

CHILD = r'''
import os, sys, json, torch
sys.path.insert(0, os.environ["REPO"])
from tutel_amd import ops
def timeit(fn, iters=40, warmup=8):
    for _ in range(warmup): fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters): fn()
    e.record(); torch.cuda.synchronize()
    return s.elapsed_time(e) * 1e3 / iters
g = torch.Generator().manual_seed(0)
dt = torch.bfloat16
E, C, M, H = 64, 128, 2048, 2048
x = torch.randn([E, C, M], generator=g).to(dt).cuda()
w1 = (torch.randn([E, H, M], generator=g) / 45).to(dt).cuda()
w2 = (torch.randn([E, H, M], generator=g) / 45).to(dt).cuda()
b = torch.randn([E, H], generator=g).to(dt).cuda()
h = ops.expert_gemm(x, w1, b, True, act="relu")
y = ops.expert_gemm(h, w2, b, False)
ref_h = torch.relu(torch.matmul(x[:2].float(), w1[:2].float().permute(0, 2, 1)) + b[:2].float().unsqueeze(1))
ref_y = torch.matmul(h[:2].float(), w2[:2].float()) + b[:2].float().unsqueeze(1)
ok = bool(((h[:2].float() - ref_h).abs() <= 2**-7 * ref_h.abs() + 2e-3).all()) and bool(((y[:2].float() - ref_y).abs() <= 2**-7 * ref_y.abs() + 2e-3).all())
r = {"ok": ok, "fc1": timeit(lambda: ops.expert_gemm(x, w1, b, True, act="relu")), "fc2": timeit(lambda: ops.expert_gemm(h, w2, b, False))}
El, R, M4 = 8, 1024, 4096
a4 = torch.randn([El, R, M4], generator=g).to(dt).cuda()
w4 = (torch.randn([El, M4, M4], generator=g) / 64).to(dt).cuda()
r["c4_fc1"] = timeit(lambda: ops.expert_gemm(a4, w4, None, True, act="relu"), iters=15, warmup=4)
r["c4_fc2"] = timeit(lambda: ops.expert_gemm(a4, w4, None, False), iters=15, warmup=4)
print(json.dumps(r))
'''


def main():
    repo = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cfgs = [int(c) for c in (sys.argv[1].split(",") if len(sys.argv) > 1 else "0,1,2,3,4,5".split(","))]
    rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    res = {c: [] for c in cfgs}
    for _ in range(rounds):
        for c in cfgs:
            env = dict(os.environ, TUTEL_AMD_GEMM_IMPL=str(c), REPO=repo)
            out = subprocess.run([sys.executable, "-c", CHILD], env=env, capture_output=True, text=True)
            line = [l for l in out.stdout.splitlines() if l.startswith("{")]
            if not line:
                print("cfg", c, "FAILED", out.stderr[-500:])
                continue
            res[c].append(json.loads(line[-1]))
    for c in cfgs:
        if res[c]:
            best = {k: min(r[k] for r in res[c]) for k in ("fc1", "fc2", "c4_fc1", "c4_fc2")}
            print("cfg %d ok=%s  fc1 %.1f us  fc2 %.1f us   c4_fc1 %.1f us  c4_fc2 %.1f us" % (
                c, all(r["ok"] for r in res[c]), best["fc1"], best["fc2"], best["c4_fc1"], best["c4_fc2"]))


if __name__ == "__main__":
    main()
