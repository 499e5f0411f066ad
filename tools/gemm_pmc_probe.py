"""Dev tool: run the two expert GEMMs at one shape a few times (for rocprofv3 --pmc passes)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tutel_amd import ops
E, R, M, H = [int(v) for v in (sys.argv[1:5] if len(sys.argv) > 4 else (8, 1024, 2048, 2048))]
g = torch.Generator().manual_seed(0)
x = torch.randn([E, R, M], generator=g).bfloat16().cuda()
w1 = (torch.randn([E, H, M], generator=g) / 45).bfloat16().cuda()
w2 = (torch.randn([E, H, M], generator=g) / 45).bfloat16().cuda()
b = torch.randn([E, H], generator=g).bfloat16().cuda()
import os
from tutel_amd import _lib
if os.environ.get("GEMM_TILE"):
    ops.set_option(_lib.OPT_GEMM_TILE, int(os.environ["GEMM_TILE"]))
for _ in range(12):
    h = ops.expert_gemm(x, w1, b, True, act="relu")
    y = ops.expert_gemm(x, w2, b, True)
torch.cuda.synchronize()
