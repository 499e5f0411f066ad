#!/usr/bin/env python3
"""Run under `rocprofv3 --kernel-trace`: a few eager forwards of the overlapped pipeline (degree 2, stages on two side streams) at
the EP-8 rank shape through the library's 1-rank RCCL communicator.  tools/r3_ep_trace_summary.py turns the trace into a timeline."""
import os
import sys

import torch
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29581", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    dev = torch.device("cuda", 0)
    dist.init_process_group("nccl", device_id=dev)
    from tutel import moe
    from tutel_amd import _lib, ops
    from tutel_amd.impls import ep_native
    T, M, H, E, k = 4096, 2048, 2048, 8, 2
    torch.set_default_dtype(torch.bfloat16)
    layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                          experts={"type": "ffn", "num_experts_per_device": E, "hidden_size_per_expert": H,
                                   "activation_fn": lambda t: torch.nn.functional.relu(t)}).cuda().eval()
    torch.set_default_dtype(torch.float32)
    x = torch.randn([T, M], device=dev).bfloat16()
    ep_native._FORCE_COMM = True
    ops.set_option(_lib.OPT_EP_STREAMS, int(os.environ.get("EP_STREAMS", "2")))
    with torch.no_grad():
        for _ in range(30):
            layer(x, a2a_ffn_overlap_degree=2)
        torch.cuda.synchronize()
    ep_native.destroy_all()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
