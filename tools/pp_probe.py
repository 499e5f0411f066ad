"""Dev tool: the 256-row-tile GEMM kernels against each other and against torch.bmm (hipBLASLt), interleaved
and repeated; bitwise comparison of the outputs.  python tools/pp_probe.py [quick]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops, _lib  # noqa: E402

VARIANTS = (("256x256", 1), ("pingpong", 4), ("256x128", 3), ("128", 0))


def main():
    quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
    g = torch.Generator().manual_seed(0)
    shapes = [(8, 1024, 2048, 2048), (8, 1024, 4096, 4096), (32, 256, 2048, 2048), (16, 512, 2048, 2048), (64, 160, 2048, 2048),
              (8, 1000, 2048, 2048), (3, 300, 512, 320)]
    for El, R, N, K in shapes:
        a = torch.randn([El, R, K], generator=g).bfloat16().cuda()
        w = (torch.randn([El, N, K], generator=g) / K ** 0.5).bfloat16().cuda()
        b = torch.randn([El, N], generator=g).bfloat16().cuda()
        outs, res = {}, {}
        for name, opt in VARIANTS:
            ops.set_option(_lib.OPT_GEMM_TILE, opt)
            outs[name] = ops.expert_gemm(a, w, b, True, act="relu").clone()
        torch.cuda.synchronize()
        same = torch.equal(outs["256x256"], outs["pingpong"])
        ref = torch.relu(torch.matmul(a.float(), w.float().transpose(1, 2)) + b.float().unsqueeze(1))
        err = float((outs["pingpong"].float() - ref).abs().max())
        wt = w.transpose(1, 2).contiguous()  # (the transposed VIEW faults inside the library at 8x1024x2048x2048)
        reps = 1 if quick else 3
        for rep in range(reps):
            for name, opt in VARIANTS + (("torch.bmm", None),):
                if opt is not None:
                    ops.set_option(_lib.OPT_GEMM_TILE, opt)
                    fn = lambda: ops.expert_gemm(a, w, b, True, act="relu")  # noqa: E731
                else:
                    fn = lambda: torch.matmul(a, wt)  # noqa: E731
                for _ in range(10):
                    fn()
                torch.cuda.synchronize()
                s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                s.record()
                for _ in range(50):
                    fn()
                e.record()
                torch.cuda.synchronize()
                res.setdefault(name, []).append(s.elapsed_time(e) * 20)
                if os.environ.get("PP_VERBOSE"):
                    print("   ", name, rep, round(res[name][-1], 1), flush=True)
        ops.set_option(_lib.OPT_GEMM_TILE, -1)
        fl = 2 * El * R * N * K
        print(f"{El}x{R}x{N}x{K}: bitwise equal {same}, max err vs fp32 {err:.3e};",
              {k: [round(v, 1) for v in vs] for k, vs in res.items()}, "TFLOP/s", {k: round(fl / min(vs) * 1e-6) for k, vs in res.items()}, flush=True)


if __name__ == "__main__":
    main()
