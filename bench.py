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
N > 1: experts are sharded E_loc = 64/N per rank (expert parallel, RCCL all-to-all over xGMI through the
library's own communicator, a2a_ffn_overlap_degree 2), every rank keeps its own 4096 tokens -> per-GPU work
is fixed: weak scaling, value = N * 4096 / t.

Timing: `--settle` untimed initialisation passes (allocator / weight pre-layout / clock state; reported in
the JSON), W untimed warm-up steps; barrier + synchronize; exactly K steps; synchronize + barrier; MAX over
ranks.  Rank 0 prints ONE JSON line.  The timed region carries NO events (a HIP event record drains the queue: 5-15 % of a
step here).  Three more passes of the same K steps follow it, bracketed the same way: one device-scope event per step
(`step_ms` min / median), events around the two expert GEMM launches on their launch stream (`roofline`), events around
every launch (`stages`) -- all live, inside this script, on the same tensors.

`roofline`: the dominant kernel is the fc1 grouped GEMM.  N = 1 (128 rows per expert):
expert_gemm_glds_kernel<bf16,k-major,relu>, HBM-bound; achieved = algorithmic bytes per launch
(E_loc*H*M weights + E_loc*R*M tokens + E_loc*R*H hidden out, x2 bytes) / its average duration.
256 rows or more per expert and launch (N > 1, --tokens 65536): expert_gemm_pp_kernel, MFMA-bound; achieved = flop
per launch / its average duration.
`cpu_baseline`: the CPU oracle (a port of the reference CPU path, oracle/moe_oracle.py) timed on this box's host
cores on a bounded sample, rank 0, N=1 only, next to the figure BASELINE.md measured with the reference itself.
Checker code is used here ONLY as that reported baseline; it is never part of the measured GPU path.
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

HBM_PEAK_GBS = 8000.0        # MI355X HBM3E spec (/opt/skills/guides/MI355X_MICROARCH.md)
HBM_ACHIEVABLE_GBS = 6290.0  # same guide: measured float4 copy (79 % of spec)
MFMA_PEAK_TFLOPS = 2500.0    # dense bf16 MFMA peak, same guide


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


class GateTimer:
    """HIP events around the gate projection (a library GEMM issued by torch, so not covered by the C-ABI stage timer)."""

    def __init__(self, gate):
        self.gate, self.real, self.events, self.on = gate, gate.forward, [], False
        gate.forward = self

    def __call__(self, x):
        if not self.on:
            return self.real(x)
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        s.record()
        out = self.real(x)
        e.record()
        self.events.append((s, e))
        return out

    def avg_us(self):
        return sum(s.elapsed_time(e) for s, e in self.events) * 1e3 / max(1, len(self.events))


def physical_cores():
    try:
        import psutil
        n = psutil.cpu_count(logical=False)
        if n:
            return int(n)
    except Exception:
        pass
    return max(1, (os.cpu_count() or 2) // 2)


def cpu_baseline(T, M, H, E, k, max_seconds=20.0):
    from oracle import moe_oracle as O
    threads = min(physical_cores(), 32)  # ATen GEMMs stop scaling (and the C scatter loops are single-threaded) well before that
    old = torch.get_num_threads()
    torch.set_num_threads(threads)
    try:
        x, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=torch.float32, seed=0)
        with torch.no_grad():
            O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k)  # warm-up (page-in, thread pool)
            times = []
            t_all = time.time()
            while len(times) < 10 and (time.time() - t_all) < max_seconds:
                t0 = time.time()
                O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k)
                times.append(time.time() - t0)
    finally:
        torch.set_num_threads(old)
    dt = sorted(times)[len(times) // 2]
    return {"value": round(T / dt, 1), "unit": "tokens/s", "cores": threads, "kind": "port",
            "kind_note": "oracle/moe_oracle.py moe_forward: the reference's CPU path restated (ATen softmax / matmul + C scatter / gather "
                         "loops); /root/reference does not exist on the GPU box, so the reference itself cannot be timed here",
            "ms_per_step": round(dt * 1e3, 2), "dtype": "f32", "host_logical_cpus": os.cpu_count(), "host_physical_cores": physical_cores(),
            "sample": f"median of {len(times)} forward passes of the same {T}-token workload (fp32), torch.set_num_threads({threads})",
            "reference_measured": {"value": "13-15 k tokens/s (0.27-0.31 s per forward)", "cores": 8, "kind": "reference",
                                   "source": "BASELINE.md section 2: the reference's own helloworld --device=cpu --eval at this shape, survey container (8 x Xeon 2.1 GHz)"}}


def run_timed(step, x, steps, world, timer_gate, mode=2, marks=True):
    """exactly `steps` steps between barrier + synchronize pairs; returns (elapsed_s, per-step ms list, stage report).
    mode 2: HIP events around the two expert GEMMs only (the timed region); mode 1: around every launch (breakdown pass)."""
    from tutel_amd import ops
    ops.stage_timing(mode)          # (host-side setup first: nothing but the clock read stands between the synchronize
    timer_gate.on = mode == 1       #  that closes the bracket and the first step -- event creation takes ~0.2 ms, long
    ops.marks_reserve(steps + 1)    #  enough for an idle GPU to drop its clocks)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if marks:
        ops.mark()
    y = None
    for i in range(steps):
        y = step(x)
        if marks:
            ops.mark()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ops.stage_timing(0)
    timer_gate.on = False
    per_step = ops.marks_report(steps)
    return t1 - t0, per_step, ops.stage_report(), y


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--settle", type=int, default=200,
                    help="untimed initialisation passes before the warm-up steps (~60 ms: allocator steady state, k-major weight "
                         "copies, the GPU leaving its idle clock state); reported in the JSON line")
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
    ap.add_argument("--no_extra", action="store_true", help="skip the short BASELINE configs[2] (dropless) measurement appended at N=1")
    ap.add_argument("--graph", action="store_true", help="replay the forward from a captured HIP graph")
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
    local_rank %= max(1, torch.cuda.device_count())   # a launcher that hands every rank ONE visible device numbers it 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        if share:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    from tutel_amd import _lib, ops
    from tutel_amd.impls import ep_native
    _lib.lib()  # fail loudly if the HIP library is missing
    T, M, H, E, k = args.tokens, args.model_dim, args.hidden_size, args.experts, args.top
    assert E % world == 0
    E_loc = E // world
    overlap = args.a2a_ffn_overlap_degree or (2 if world > 1 else 1)
    dtype = getattr(torch, args.dtype)
    dname = "bf16" if dtype == torch.bfloat16 else "fp16"
    es = 2

    # like helloworld.py:77,93-94: the capacity factor belongs to the gate, megablocks_size to the forward call
    layer = build_layer(M, H, E_loc, k, rank, overlap, dtype, args.fp32_gate, args.capacity_factor).to(dev).eval()
    torch.manual_seed(0)  # same tokens on every rank, like helloworld.py:112-113
    x = torch.randn([16, T // 16, M], dtype=torch.float32).to(dtype).to(dev)
    gate_timer = GateTimer(layer.gates[0])
    fwd_kw = dict(megablocks_size=args.megablocks_size) if args.megablocks_size else {}
    step = (lambda t: layer(t, **fwd_kw)) if fwd_kw else layer
    launch = "eager"
    if args.graph:
        from tutel_amd.impls.graph import GraphedForward
        step = GraphedForward(layer, x, **fwd_kw)  # same kernels (and collectives), enqueued by one hipGraphLaunch per step
        launch = "hip-graph replay"

    with torch.no_grad():
        for _ in range(args.settle):
            step(x)
        for _ in range(args.warmup):
            y = step(x)
        # THE timed region: exactly K steps, nothing else on the stream.  HIP events are NOT free on this part: a record makes
        # the queue drain and release before it takes its timestamp (measured, profiles/r02_event_overhead.txt: 0.2605 ms/step
        # bare, 0.263 with one mark per step, 0.275 with events around the two GEMMs, 0.305 around every launch), so the
        # event-based numbers come from three more passes of the same steps right after it, each bracketed the same way:
        elapsed, _, _, y = run_timed(step, x, args.steps, world, gate_timer, mode=0, marks=False)
        nb = args.steps
        eager = (lambda t: layer(t, **fwd_kw))
        _, per_step, _, _ = run_timed(step, x, nb, world, gate_timer, mode=0, marks=True)     # one mark per step: min / median
        run_timed(eager, x, 3, 1, gate_timer, mode=2, marks=False)                            # (fills the event pool)
        _, _, gemms, _ = run_timed(eager, x, nb, world, gate_timer, mode=2, marks=False)      # events around fc1 / fc2: roofline
        run_timed(eager, x, 3, 1, gate_timer, mode=1, marks=False)
        gate_timer.events.clear()
        _, _, stages, _ = run_timed(eager, x, nb, world, gate_timer, mode=1, marks=False)     # events around every launch: stages
    if world > 1:
        tt = torch.tensor([elapsed], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt)
    assert torch.isfinite(y.float()).all()

    C = int(layer.protected_shape[1]) // world  # capacity the layer actually used (dropless: max expert load)
    R = world * C                               # rows per local expert
    stage_us = {name: round(tot / nb, 2) for name, (tot, cnt) in stages.items() if cnt}
    launches = {name: cnt for name, (tot, cnt) in stages.items() if cnt}
    stage_us["gate_projection(hipBLASLt)"] = round(gate_timer.avg_us(), 2)
    fc1_tot, fc1_n = gemms["expert_fc1"]
    fc2_tot, fc2_n = gemms["expert_fc2"]
    fc1_us, fc2_us = fc1_tot / max(1, fc1_n), fc2_tot / max(1, fc2_n)   # per LAUNCH (one pipeline stage when N > 1)
    rows_per_launch, experts_per_launch = R, E_loc
    if world > 1 and overlap > 1:
        pl = ep_native.plan(E, world, C, overlap)
        rows_per_launch, experts_per_launch = pl["gemm_rows"], pl["experts_per_stage"]
    gemm_bytes = (experts_per_launch * H * M + experts_per_launch * rows_per_launch * (M + H)) * es
    gemm_flops = 2 * experts_per_launch * rows_per_launch * M * H
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and world == 1 and (T, M, H, E, k) == (4096, 2048, 2048, 64, 2) and args.capacity_factor == 1.0:
        traffic = json.load(open(tpath)).get("expert_gemm_fc1_hbm_bytes_per_launch")

    fc2_obj = {"avg_launch_us": round(fc2_us, 2), "achieved_GBs": round(gemm_bytes / max(fc2_us, 1e-9) * 1e-3, 1),
               "tflops": round(gemm_flops / max(fc2_us, 1e-9) * 1e-6, 1), "launches_timed": fc2_n}
    if rows_per_launch >= 256:
        tf = gemm_flops / fc1_us * 1e-6
        roofline = {"bound": "mfma", "kernel": f"expert_gemm_pp_kernel<{dname},relu> (fc1 grouped GEMM; 256x256 ping-pong tile, or 256x128 / 128x128 when the grid is small)",
                    "achieved": round(tf, 1), "peak": MFMA_PEAK_TFLOPS, "unit": "TFLOP/s", "frac": round(tf / MFMA_PEAK_TFLOPS, 4),
                    "traffic": None, "flops_per_launch": gemm_flops, "rows_per_expert": rows_per_launch,
                    "avg_launch_us": round(fc1_us, 2), "launches_timed": fc1_n, "fc2_gemm": fc2_obj}
    else:
        gbs = gemm_bytes / fc1_us * 1e-3
        roofline = {"bound": "hbm", "kernel": f"expert_gemm_glds_kernel<{dname},k-major,relu> (fc1 grouped GEMM, LDS-DMA)",
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 4), "achievable_GBs": HBM_ACHIEVABLE_GBS, "traffic": traffic,
                    "algorithmic_bytes_per_launch": gemm_bytes, "avg_launch_us": round(fc1_us, 2), "launches_timed": fc1_n,
                    "fc2_gemm": fc2_obj, "mfma_tflops_fc1": round(gemm_flops / fc1_us * 1e-6, 1)}

    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * T / (elapsed / args.steps)
        # algorithmic bytes of the whole forward on one rank: gate reads x; encode (T + E*C) rows; FFN weights + rows in/out;
        # decode (n_kept + T) rows; all-to-all bytes are link traffic, not HBM, and are listed separately
        layer_bytes = (T * M + (T + E * C) * M + 2 * E_loc * H * M + 2 * E_loc * R * (M + H) + (min(k * T, E * C) + T) * M) * es
        srt = sorted(per_step)
        out = {
            "metric": "MoE-layer fwd tokens/sec, 4096 tok x H=2048 x E=64 top-2",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle": args.settle,
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dname, "data": "synthetic" if not share else "synthetic; TEST HOOK: all ranks share one GPU, host-staged all-to-all -- not a measurement",
            "step_ms": {"mean_wall": round(ms, 4), "min": round(srt[0], 4), "median": round(srt[len(srt) // 2], 4), "max": round(srt[-1], 4),
                        "per_step": [round(v, 4) for v in per_step],
                        "note": "mean_wall = the timed region / steps (no events inside it); min / median / max / per_step from a second pass "
                                "of the same steps with one device-scope HIP event per step (rank 0); the first step after the "
                                "synchronize is host-bound (Python + first launches on an idle queue)"},
            "timing_passes": "1: K steps, wall clock only (value). 2: K steps + one mark per step (step_ms). 3: K steps + HIP events "
                             "around the two expert GEMM launches on their launch stream (roofline). 4: K steps + events around every "
                             "launch (stages). Each pass is bracketed by synchronize (+ barrier) like the first; events cost 5-15 % "
                             "of a step on this GPU, which is why the timed region carries none.",
            "config": {"workload": "BASELINE.json configs[1]: tutel.moe.moe_layer forward (eval), per GPU 4096 tokens "
                                   "(batch 16 x 256) x model_dim 2048, hidden 2048, 64 global experts, top-2, "
                                   f"capacity_factor {args.capacity_factor}, ReLU, {dname}, random-init weights",
                       "tokens_per_gpu": T, "model_dim": M, "hidden_size": H, "global_experts": E, "top_k": k,
                       "capacity": C, "parallelism": f"ep{world}" if world > 1 else "single-gpu",
                       "a2a_ffn_overlap_degree": overlap, "fp32_gate": bool(args.fp32_gate),
                       "capacity_factor": args.capacity_factor, "megablocks_size": args.megablocks_size, "launch": launch,
                       "exchange": ("library RCCL communicator (tutel_amd_ep_forward)" if ep_native._comms and any(ep_native._comms.values())
                                    else "torch.distributed all_to_all_single") if world > 1 else "none (single rank)"},
            "roofline": roofline,
            "stages": {"avg_us_per_step": stage_us, "launches_timed": launches,
                       "sum_us": round(sum(stage_us.values()), 2),
                       "steps": nb,
                       "note": "pass 4: HIP events around EVERY launch (tutel_amd_stage_timing(1)); the sum exceeds a bare step because "
                               "every event record drains the queue"},
            "layer_roofline": {"algorithmic_bytes_per_step": layer_bytes, "achieved_GBs": round(layer_bytes / (ms * 1e-3) * 1e-9, 1),
                               "frac_of_hbm_peak": round(layer_bytes / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4),
                               "frac_of_hbm_achievable": round(layer_bytes / (ms * 1e-3) * 1e-9 / HBM_ACHIEVABLE_GBS, 4)},
        }
        if world == 1 and not args.no_extra and args.capacity_factor == 1.0 and not args.graph and (T, M, H, E, k) == (4096, 2048, 2048, 64, 2):
            # BASELINE configs[2] (same shape, dropless + megablocks): a short secondary measurement, recorded next to the headline one
            del layer, step
            lay2 = build_layer(M, H, E_loc, k, rank, 1, dtype, args.fp32_gate, 0.0).to(dev).eval()
            gt2 = GateTimer(lay2.gates[0])
            with torch.no_grad():
                for _ in range(30):
                    lay2(x, megablocks_size=4)
                el2, _, _, _ = run_timed(lambda t: lay2(t, megablocks_size=4), x, 30, 1, gt2, mode=0, marks=False)
                _, _, st2, _ = run_timed(lambda t: lay2(t, megablocks_size=4), x, 30, 1, gt2, mode=2, marks=False)
            ms2 = el2 / 30 * 1e3
            out["extra"] = {"dropless_configs2": {
                "workload": "BASELINE.json configs[2]: same shape, capacity_factor 0 (capacity = max expert load, read back each step), megablocks_size 4",
                "value": round(T / (el2 / 30), 1), "unit": "tokens/s", "ms_per_step": round(ms2, 4), "steps": 30,
                "capacity": int(lay2.protected_shape[1]),
                "fc1_avg_us": round(st2["expert_fc1"][0] / max(1, st2["expert_fc1"][1]), 2),
                "fc2_avg_us": round(st2["expert_fc2"][0] / max(1, st2["expert_fc2"][1]), 2)}}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, M, H, E, k)
        print(json.dumps(out), flush=True)
    if world > 1:
        ep_native.destroy_all()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
