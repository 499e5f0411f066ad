#!/usr/bin/env python3
"""bench.py -- MoE-layer forward throughput on MI355X (BASELINE.json metric).

    python bench.py --gpus 1 --steps K --warmup W
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
           --master-port P bench.py --gpus N --steps K --warmup W

A "step" = one forward of tutel.moe.moe_layer (the drop-in API, HIP kernels underneath) under
torch.no_grad() on one batch of synthetic tokens already resident in HBM -- the semantics of the
reference's `helloworld.py --eval` (helloworld.py:141-146).

Workload (BASELINE.json configs[1], the configuration the metric is quoted on):
    per GPU: 4096 tokens (batch 16 x 256) x model_dim 2048, hidden 2048, 64 GLOBAL experts,
    top-2, capacity_factor 1.0 (capacity 128/expert/rank), bf16, ReLU, biases on.
N > 1: experts are sharded E_loc = 64/N per rank (expert parallel, RCCL all_to_all_single over
xGMI), every rank keeps its own 4096 tokens -> per-GPU work is fixed: weak scaling,
value = N * 4096 / t.

Timing: untimed initialisation passes (--settle: allocator / weight pre-layout / clock state), W untimed
warm-up steps; barrier + synchronize; exactly K steps; synchronize + barrier; MAX over ranks.  Rank 0
prints ONE JSON line.

`roofline`: the dominant kernel is the fc1 grouped GEMM.  N = 1 (128 rows per expert):
expert_gemm_glds_kernel<bf16,k-major,relu>, HBM-bound; achieved = algorithmic bytes per launch
(E_loc*H*M weights + E_loc*R*M tokens + E_loc*R*H hidden out, x2 bytes) / its average duration.
More than 128 rows per expert and launch (N > 1, --tokens 65536, dropless): expert_gemm_big_kernel,
MFMA-bound; achieved = flop per launch / its average duration.  Durations are measured with HIP events
on the launch stream inside the timed region.
`cpu_baseline`: the CPU oracle (a port of the reference CPU path, oracle/moe_oracle.py) timed on
this box's host cores on a bounded sample, rank 0, N=1 only.  Checker code is used here ONLY as
that reported baseline; it is never part of the measured GPU path.
"""
import argparse
import json
import os
import sys
import time

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
MFMA_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA peak, same guide


def build_layer(M, H, E_loc, k, rank, overlap, dtype, fp32_gate, capacity_factor=1.0):
    from tutel import moe
    torch.set_default_dtype(dtype)
    try:
        layer = moe.moe_layer(
            gate_type={"type": "top", "k": k, "fp32_gate": fp32_gate, "capacity_factor": capacity_factor},
            experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                     "activation_fn": lambda t: torch.nn.functional.relu(t)},
            model_dim=M, seeds=(1, rank + 1, 1), a2a_ffn_overlap_degree=overlap)
    finally:
        torch.set_default_dtype(torch.float32)
    return layer


class GemmTimer:
    """Wraps tutel_amd.ops.expert_gemm: HIP events around every launch of the dominant kernel."""

    def __init__(self):
        from tutel_amd import ops
        self.ops, self.real, self.events, self.on = ops, ops.expert_gemm, {True: [], False: []}, False
        self.bytes, self.flops = {True: 0, False: 0}, {True: 0, False: 0}  # algorithmic, per launch (last seen)
        self.rows = {True: 0, False: 0}                                    # rows per expert of the last launch
        ops.expert_gemm = self
        self.real_gather = ops.expert_gemm_gather
        ops.expert_gemm_gather = self.gather

    def __call__(self, a, w, bias, w_kmajor, *args, **kw):
        if not self.on:
            return self.real(a, w, bias, w_kmajor, *args, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = self.real(a, w, bias, w_kmajor, *args, **kw)
        e.record()
        km = kw.get("act", "none") != "none"  # True: fc1 (bias + ReLU fused), False: fc2
        self.events[km].append((s, e))
        E_loc, N, K = (w.shape[0], w.shape[1], w.shape[2]) if w_kmajor else (w.shape[0], w.shape[2], w.shape[1])
        R = kw.get("R") or a.shape[1]
        self.rows[km] = R
        self.bytes[km] = (w.numel() + E_loc * R * K + E_loc * R * N) * w.element_size()
        self.flops[km] = 2 * E_loc * R * N * K
        return out

    def gather(self, x, smap, w, bias, w_kmajor, act, R, **kw):
        """fc1 with fast_encode fused (single rank): same kernel, rows gathered from the tokens.  Algorithmic
        bytes are counted exactly as for the plain fc1 launch (weights + E*R token rows + hidden out)."""
        if not self.on:
            return self.real_gather(x, smap, w, bias, w_kmajor, act, R, **kw)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = self.real_gather(x, smap, w, bias, w_kmajor, act, R, **kw)
        e.record()
        km = True  # the gathered launch is always fc1
        self.rows[km] = R
        self.events[km].append((s, e))
        E_loc, N, K = (w.shape[0], w.shape[1], w.shape[2]) if w_kmajor else (w.shape[0], w.shape[2], w.shape[1])
        self.bytes[km] = (w.numel() + E_loc * R * K + E_loc * R * N) * w.element_size()
        self.flops[km] = 2 * E_loc * R * N * K
        return out

    def avg_us(self, fc1):
        ev = self.events[fc1]
        return sum(s.elapsed_time(e) for s, e in ev) * 1e3 / max(1, len(ev)), len(ev)


def cpu_baseline(T, M, H, E, k, max_seconds=25.0):
    from oracle import moe_oracle as O
    x, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=torch.float32, seed=0)
    with torch.no_grad():
        O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k)  # warm-up (page-in, thread pool)
        t0, n = time.time(), 0
        while n < 10 and (time.time() - t0) < max_seconds:
            O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k)
            n += 1
        dt = (time.time() - t0) / max(1, n)
    return {"value": round(T / dt, 1), "unit": "tokens/s", "cores": torch.get_num_threads(), "kind": "port",
            "ms_per_step": round(dt * 1e3, 2), "dtype": "f32",
            "sample": f"{n} forward passes of the same {T}-token workload (fp32, oracle/moe_oracle.py moe_forward = "
                      f"the reference's CPU path restated: ATen softmax/matmul + C scatter/gather loops), host cores only"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=1500, help="untimed initialisation passes before the warm-up steps")
    ap.add_argument("--tokens", type=int, default=4096)
    ap.add_argument("--model_dim", type=int, default=2048)
    ap.add_argument("--hidden_size", type=int, default=2048)
    ap.add_argument("--experts", type=int, default=64, help="GLOBAL expert count")
    ap.add_argument("--top", type=int, default=2)
    ap.add_argument("--a2a_ffn_overlap_degree", type=int, default=None)
    ap.add_argument("--fp32_gate", action="store_true")
    ap.add_argument("--dtype", choices=["bfloat16", "float16"], default="bfloat16", help="the headline metric is bf16")
    ap.add_argument("--capacity_factor", type=float, default=1.0,
                    help="BASELINE configs[2]: 0 = dropless (capacity read back from the device each step)")
    ap.add_argument("--megablocks_size", type=int, default=0, help="configs[2]: row granularity of the dropless expert GEMMs")
    ap.add_argument("--no_cpu_baseline", action="store_true")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured HIP graph (N=1 only)")
    args = ap.parse_args()

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # Test hook (not a measurement mode): TUTEL_AMD_BENCH_SHARE_GPU=1 runs every rank on cuda:0 with a gloo
    # rendezvous and a host-staged all-to-all, so the N > 1 code path of this script can be exercised on a
    # single-GPU box.  The driver's multi-GPU runs use one GPU per rank over RCCL.
    share = os.environ.get("TUTEL_AMD_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from tutel_amd import _lib
    _lib.lib()  # fail loudly if the HIP library is missing
    T, M, H, E, k = args.tokens, args.model_dim, args.hidden_size, args.experts, args.top
    assert E % world == 0
    E_loc = E // world
    overlap = args.a2a_ffn_overlap_degree or (2 if world > 1 else 1)
    dtype = getattr(torch, args.dtype)
    dname = "bf16" if dtype == torch.bfloat16 else "fp16"

    # like helloworld.py:77,93-94: the capacity factor belongs to the gate, megablocks_size to the forward call
    layer = build_layer(M, H, E_loc, k, rank, overlap, dtype, args.fp32_gate, args.capacity_factor).to(dev).eval()
    torch.manual_seed(0)  # same tokens on every rank, like helloworld.py:112-113
    x = torch.randn([16, T // 16, M], dtype=torch.float32).to(dtype).to(dev)
    timer = GemmTimer()
    fwd_kw = {}
    if args.megablocks_size:
        fwd_kw = dict(megablocks_size=args.megablocks_size)
    step = (lambda t: layer(t, **fwd_kw)) if fwd_kw else layer
    if args.graph and world == 1:
        from tutel_amd.impls.graph import GraphedForward
        step = GraphedForward(layer, x)  # same kernels, enqueued by one hipGraphLaunch per step

    with torch.no_grad():
        # initialisation passes before the W warm-up steps of the contract: the caching allocator reaches its
        # steady state, the k-major weight copies are laid out, and the GPU leaves its idle clock state (the
        # first few hundred ms after start-up run 5-8 % slower: 20 timed steps measured 0.313 ms without, 0.283 with); never timed
        for _ in range(args.settle):
            step(x)
        for _ in range(args.warmup):
            y = step(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        timer.on = not (args.graph and world == 1)  # events cannot be recorded inside a replayed graph
        t0 = time.perf_counter()
        for _ in range(args.steps):
            y = step(x)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        timer.on = False
    elapsed = t1 - t0
    if world > 1:
        tt = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    assert torch.isfinite(y.float()).all()

    C = int(layer.protected_shape[1]) // world  # capacity the layer actually used (dropless: max expert load)
    if not timer.events[True]:  # graph mode: time the two GEMM launches in a short eager pass after the timed region
        timer.on = True
        with torch.no_grad():
            for _ in range(20):
                step(x)
        torch.cuda.synchronize()
        timer.on = False
    fc1_us, n1 = timer.avg_us(True)
    fc2_us, n2 = timer.avg_us(False)
    fc1_bytes, fc2_bytes = timer.bytes[True], timer.bytes[False]  # per LAUNCH (one overlap chunk when N > 1)
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1 and (T, M, H, E, k) == (4096, 2048, 2048, 64, 2):
        traffic = json.load(open(tpath)).get("expert_gemm_fc1_hbm_bytes_per_launch")

    fc2_obj = {"avg_launch_us": round(fc2_us, 2), "achieved_GBs": round(fc2_bytes / max(fc2_us, 1e-9) * 1e-3, 1),
               "tflops": round(timer.flops[False] / max(fc2_us, 1e-9) * 1e-6, 1), "launches_timed": n2}
    if timer.rows[True] >= 256:
        # >= 256 rows per expert and launch (expert-parallel ranks): the 256 x 256-tile kernel, bound by the MFMA rate
        tf = timer.flops[True] / fc1_us * 1e-6
        roofline = {"bound": "mfma", "kernel": f"expert_gemm_big_kernel<{dname},k-major,relu> (fc1 grouped GEMM, 256-row tile: 256x256 or 256x128 by grid size, LDS-DMA)",
                    "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                    "traffic": None, "flops_per_launch": timer.flops[True], "rows_per_expert": timer.rows[True],
                    "avg_launch_us": round(fc1_us, 2), "launches_timed": n1, "fc2_gemm": fc2_obj}
    else:
        roofline = {"bound": "hbm", "kernel": f"expert_gemm_glds_kernel<{dname},k-major,relu> (fc1 grouped GEMM, LDS-DMA)",
                    "achieved": round(fc1_bytes / fc1_us * 1e-3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": round(fc1_bytes / fc1_us * 1e-3 / HBM_PEAK_GBS, 4), "traffic": traffic,
                    "algorithmic_bytes_per_launch": fc1_bytes, "avg_launch_us": round(fc1_us, 2), "launches_timed": n1,
                    "fc2_gemm": fc2_obj, "mfma_tflops_fc1": round(timer.flops[True] / fc1_us * 1e-6, 1)}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * T / (elapsed / args.steps)
        out = {
            "metric": "MoE-layer fwd tokens/sec, 4096 tok x H=2048 x E=64 top-2",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dname, "data": "synthetic" if not share else "synthetic; TEST HOOK: all ranks share one GPU, host-staged all-to-all -- not a measurement",
            "config": {"workload": "BASELINE.json configs[1]: tutel.moe.moe_layer forward (eval), per GPU 4096 tokens "
                                   "(batch 16 x 256) x model_dim 2048, hidden 2048, 64 global experts, top-2, "
                                   f"capacity_factor {args.capacity_factor}, ReLU, {dname}, random-init weights",
                       "tokens_per_gpu": T, "model_dim": M, "hidden_size": H, "global_experts": E, "top_k": k,
                       "capacity": C, "parallelism": f"ep{world}" if world > 1 else "single-gpu",
                       "a2a_ffn_overlap_degree": overlap, "fp32_gate": bool(args.fp32_gate),
                       "capacity_factor": args.capacity_factor, "megablocks_size": args.megablocks_size,
                       "launch": "hip-graph replay" if (args.graph and world == 1) else "eager"},
            "roofline": roofline,
        }
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, M, H, E, k)
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
