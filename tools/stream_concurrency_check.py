#!/usr/bin/env python3
"""Do kernels on two HIP streams of one process run CONCURRENTLY on this box?  Two CU-pinning kernels (tools/scratch/cu_pin.hip,
16 blocks x 1 ms each, start / end device wall-clock stamps per block) on two streams; prints how much their intervals overlap
for several stream pairings, plus the environment variables that influence queue scheduling."""
import ctypes
import json
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from tools.contention_probe import pin_lib  # noqa: E402


def main():
    torch.cuda.set_device(0)
    P = pin_lib()
    a = torch.zeros([64], dtype=torch.int64, device="cuda")
    b = torch.zeros([64], dtype=torch.int64, device="cuda")
    out = {"env": {k: v for k, v in os.environ.items() if k.split("_")[0] in ("HIP", "AMD", "GPU", "HSA", "ROC", "ROCR", "NCCL", "RCCL")}}

    def raw(s):
        return s.cuda_stream

    def make_raw_nonblocking():
        hip = ctypes.CDLL("libamdhip64.so")
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithFlags(ctypes.byref(st), 1) == 0   # hipStreamNonBlocking
        return st.value
    if os.environ.get("CHECK_WITH_RCCL"):   # does an initialised RCCL communicator (its own queues) change the picture?
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29578", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
        import torch.distributed as dist
        dist.init_process_group("nccl", device_id=torch.device("cuda", 0))
        from tutel_amd.impls import ep_native
        comm = ep_native.communicator(None, torch.device("cuda", 0))
        t = torch.zeros([1 << 20], device="cuda")
        comm.all_to_all(torch.empty_like(t), t)
        torch.cuda.synchronize()
        out["rccl"] = "process group + library communicator initialised, one all-to-all done"
    pairs = {"torch side stream + torch side stream": (raw(torch.cuda.Stream()), raw(torch.cuda.Stream())),
             "default (null) stream + torch side stream": (raw(torch.cuda.default_stream()), raw(torch.cuda.Stream())),
             "raw hipStreamNonBlocking + raw hipStreamNonBlocking": (make_raw_nonblocking(), make_raw_nonblocking()),
             "default (null) stream + raw hipStreamNonBlocking": (raw(torch.cuda.default_stream()), make_raw_nonblocking()),
             "torch side stream + default (null) stream": (raw(torch.cuda.Stream()), raw(torch.cuda.default_stream())),
             "raw hipStreamNonBlocking + default (null) stream": (make_raw_nonblocking(), raw(torch.cuda.default_stream()))}
    # the library's side stream against the caller's default stream, many fresh streams: how often do the two serialise (streams that
    # share a hardware queue do), for a normal-priority and for a high-priority side stream?
    hip = ctypes.CDLL("libamdhip64.so")
    lo_p, hi_p = ctypes.c_int(), ctypes.c_int()
    hip.hipDeviceGetStreamPriorityRange(ctypes.byref(lo_p), ctypes.byref(hi_p))
    out["priority_range(least, greatest)"] = [lo_p.value, hi_p.value]

    def make_prio(p):
        st = ctypes.c_void_p()
        assert hip.hipStreamCreateWithPriority(ctypes.byref(st), 1, p) == 0
        return st.value
    null = raw(torch.cuda.default_stream())
    for label, mk in (("normal-priority side stream", make_raw_nonblocking), ("high-priority side stream", lambda: make_prio(hi_p.value))):
        res = []
        for trial in range(8):
            sd = mk()
            for order in ("side first", "null first"):
                torch.cuda.synchronize()
                first, second = (sd, null) if order == "side first" else (null, sd)
                assert P.cu_pin(16, 300.0, a.data_ptr(), first) == 0
                assert P.cu_pin(16, 300.0, b.data_ptr(), second) == 0
                torch.cuda.synchronize()
                ia, ib = a[:32].cpu().view(16, 2), b[:32].cpu().view(16, 2)
                lo, hi = max(int(ia[:, 0].min()), int(ib[:, 0].min())), min(int(ia[:, 1].max()), int(ib[:, 1].max()))
                res.append(round(max(0, hi - lo) / 100.0))
        out[label + ": overlap_us of 300 over 8 fresh streams x (side first, null first)"] = res
    for name, (s1, s2) in pairs.items():
        torch.cuda.synchronize()
        assert P.cu_pin(16, 1000.0, a.data_ptr(), s1) == 0
        assert P.cu_pin(16, 1000.0, b.data_ptr(), s2) == 0
        torch.cuda.synchronize()
        ia, ib = a[:32].cpu().view(16, 2), b[:32].cpu().view(16, 2)
        lo, hi = max(int(ia[:, 0].min()), int(ib[:, 0].min())), min(int(ia[:, 1].max()), int(ib[:, 1].max()))
        out[name] = {"kernel_1_us": (int(ia[:, 1].max()) - int(ia[:, 0].min())) / 100.0, "kernel_2_us": (int(ib[:, 1].max()) - int(ib[:, 0].min())) / 100.0,
                     "overlap_us": max(0, hi - lo) / 100.0, "second_started_after_first_started_us": (int(ib[:, 0].min()) - int(ia[:, 0].min())) / 100.0}
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
