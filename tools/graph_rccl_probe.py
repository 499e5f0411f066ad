"""Dev probe: can the native expert-parallel pipeline (own RCCL communicator) be captured in a HIP graph?
Each variant runs in its own process (a failing capture tends to take the process down)."""
import os
import subprocess
import sys

CODE = r"""
import os, sys, torch
sys.path.insert(0, os.getcwd())
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29543", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
import torch.distributed as dist
dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
from tutel import moe
from tutel_amd.impls import ep_native
mode, degree = sys.argv[1], int(sys.argv[2])
torch.manual_seed(0)
torch.set_default_dtype(torch.bfloat16)
layer = moe.moe_layer(gate_type={"type": "top", "k": 2, "fp32_gate": True},
                      experts={"type": "ffn", "num_experts_per_device": 8, "hidden_size_per_expert": 512,
                               "activation_fn": lambda t: torch.nn.functional.relu(t)}, model_dim=256).cuda().eval()
torch.set_default_dtype(torch.float32)
x = torch.randn(1536, 256).bfloat16().cuda()
ep_native._FORCE_COMM = True
with torch.no_grad():
    want = layer(x, a2a_ffn_overlap_degree=degree).clone()
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3):
            layer(x, a2a_ffn_overlap_degree=degree)
        torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=s, capture_error_mode=mode):
            out = layer(x, a2a_ffn_overlap_degree=degree)
    torch.cuda.current_stream().wait_stream(s)
    for _ in range(3):
        g.replay()
    torch.cuda.synchronize()
    print("CAPTURE_OK", mode, degree, bool(torch.equal(out, want)), flush=True)
os._exit(0)
"""

for mode, degree in (("global", 1), ("thread_local", 1), ("relaxed", 1), ("relaxed", 2), ("thread_local", 2)):
    r = subprocess.run([sys.executable, "-c", CODE, mode, str(degree)], capture_output=True, text=True, timeout=300)
    ok = [ln for ln in r.stdout.splitlines() if ln.startswith("CAPTURE_OK")]
    print(mode, degree, "->", ok[0] if ok else f"FAILED rc={r.returncode}: " + (r.stderr.strip().splitlines() or ["?"])[-1][:200], flush=True)
