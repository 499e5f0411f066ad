#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace` through torch.distributed.run with 2 ranks sharing cuda:0: a few eager forwards of the
expert-parallel pipeline over the IPC transport at the EP-8 rank shape (8 local experts x 1024 rows, M = H = 2048), degree from
$DEGREE.  tools/r4_ipc_trace_summary.py turns the per-process traces into one timeline of the last forward.

    rocprofv3 --kernel-trace --output-format csv -d OUT -- python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
        --master-addr 127.0.0.1 --master-port 29541 tools/r4_ipc_trace.py"""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    from tutel import moe
    from tutel_amd import _lib, ops
    from tutel_amd.impls import ep_native as EN
    EN.HOSTED, EN.TRANSPORT = False, "ipc"
    degree = int(os.environ.get("DEGREE", "2"))
    if "EP_STREAMS" in os.environ:
        ops.set_option(_lib.OPT_EP_STREAMS, int(os.environ["EP_STREAMS"]))
    T, M, H, E_loc, k = 4096, 2048, 2048, 8, 2
    torch.set_default_dtype(torch.bfloat16)
    layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                   "activation_fn": lambda t: torch.nn.functional.relu(t)}, seeds=(1, rank + 1, 1)).cuda().eval()
    torch.set_default_dtype(torch.float32)
    torch.manual_seed(rank)
    x = torch.randn([T, M], device="cuda").bfloat16()
    with torch.no_grad():
        for _ in range(12):
            layer(x, a2a_ffn_overlap_degree=degree)
        torch.cuda.synchronize()
        dist.barrier()
        for _ in range(6):       # the forwards the summary looks at: both ranks start together
            layer(x, a2a_ffn_overlap_degree=degree)
        torch.cuda.synchronize()
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
