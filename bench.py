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
N > 1: experts are sharded E_loc = 64/N per rank (expert parallel), every rank keeps its own 4096 tokens -> per-GPU work
is fixed: weak scaling, value = N * 4096 / t.  `python bench.py --gpus N` WITHOUT a launcher starts itself under
torch.distributed.run (one process per GPU); under a launcher it takes RANK / WORLD_SIZE from the environment.
Before anything is timed at N > 1 the script (round 5):
  * computes reference outputs of three DIFFERENT batches over the torch.distributed path (impls/overlap.py / communicate.py:
    torch's all_to_all_single on the process group -- no oracle here);
  * runs every exchange the library has -- "ipc" (peer stores over xGMI from fast_encode and the fc2 epilogue, flag kernels,
    epoch canaries, no collective), "rccl" (ncclAllToAll on the library's own communicator), "torch" (all_to_all_single on
    torch's communicator, two HIP streams: north_star's literal configuration) -- at a2a_ffn_overlap_degree 2 and 1: each mode
    is first a PARITY CANARY (the three batches through that mode, every rank compares with the reference, the verdict is
    all-reduced) and only then a short timed run; `ep_modes` in the line carries all of them;
  * `value` is a2a_ffn_overlap_degree 2 (north_star) on the first transport of ipc -> rccl -> torch that is available AND
    passed its canary -- a transport that fails is reported loudly (stderr + `parity.fell_back_from`) and never timed as
    the value; `best` names the fastest mode that passed.  The timed path (the HIP-graph replay when the transport allows
    one) takes the three batches once more before the timed region (`parity`).

Timing: `--settle` untimed initialisation passes (allocator / weight pre-layout / clock state; reported in
the JSON), W untimed warm-up steps; barrier + synchronize; exactly K steps; synchronize + barrier; MAX over
ranks.  Rank 0 prints ONE JSON line.  The timed region carries NO events (a HIP event record drains the queue: 5-15 % of a
step here).  Three more passes of the same K steps follow it, bracketed the same way: one device-scope event per step
(`step_ms` min / median), events around the two expert GEMM launches on their launch stream (`roofline`), events around
every launch (`stages`) -- all live, inside this script, on the same tensors.

`roofline`: the dominant kernel is the fc1 grouped GEMM.  N = 1 (128 rows per expert):
expert_gemm_big_kernel<bf16,k-major,relu,128 x 256 tile on a three-slot LDS-DMA ring> (round 4; rounds 1-3:
expert_gemm_glds_kernel, 128 x 128) -- since round 5 its FL form, which also does fast_encode's row gather and the location step
(no compute_location launch; `extra.feature_ab_ms_per_step` switches that, the scalar slot-map lookups and the split-K gate
projection off one at a time) --, HBM-bound; achieved = algorithmic bytes per launch
(E_loc*H*M weights + E_loc*R*M tokens + E_loc*R*H hidden out, x2 bytes) / its average duration.
256 rows or more per expert and launch (N > 1, --tokens 65536): expert_gemm_pp_kernel, MFMA-bound; achieved = flop
per launch / its average duration.
`roofline.traffic` / `roofline.frac_rocprof` come from profiles/traffic.json (PMC FETCH_SIZE / WRITE_SIZE passes and the rocprofv3
kernel-trace average of the same command, written by tools/profile_r05.sh) and are emitted only while the sha256 of
csrc/expert_gemm.hip equals the one stamped there -- otherwise null with the reason.  `decode` and `extra.ep8_rank_gemms` are
roofline objects of the second / third kernels of interest (fast_decode; the grouped GEMM at the per-rank shapes of an 8-way
expert-parallel run: one pipeline stage, and the whole rank).
Launch mode: with capacity_factor > 0 the forward never talks to the host, so the default is to capture it once in a HIP graph
(tutel_amd.impls.graph.GraphedForward) and REPLAY it per step -- as in round 3; `--eager` times the Python-enqueued forward
instead (GPU-paced as well: 0.09 ms of host enqueue against 0.25 ms of device work; it pays its first forward after the
synchronize at host pace), and the line always carries the other mode beside the timed one
(`launch_modes`, and `value_eager` at the top level).  If capture fails the script says so on stderr and in the line and runs eager.
N > 1: the graph is replayed as well when the exchange is the IPC transport (plain kernels and events, epochs counted on the device:
2000 replays in tests/test_ep_ipc_one_gpu.py; host cost 0.05 ms per forward against 0.16 ms eager); with RCCL on the path the
forward is timed eager, because replaying captured RCCL collectives was seen to hang after a few hundred replays with this RCCL
build (profiles/r03_ep_streams.txt) and GraphedForward refuses to capture them.
`cpu_baseline`: the reference's CPU path timed on this box's host cores on a bounded sample, rank 0, N=1 only -- fast_encode /
fast_decode through the reference's OWN compiled CPU kernels (oracle/_ref, kind "reference"; see cpu_baseline()), the plain-C port
beside it -- next to the figure BASELINE.md measured with the reference's Python itself.
Checker code is used here ONLY as that reported baseline; it is never part of the measured GPU path.
"""
import argparse
import json
import os
import socket
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


def source_sha():
    """sha256 of the grouped-GEMM kernel source the library is built from (stamps profiles/traffic.json)"""
    import hashlib
    h = hashlib.sha256()
    for f in ("expert_gemm.hip", "gemm_dev.h", "common.h"):   # (round 6: the ring tile lives in gemm_dev.h)
        h.update(open(os.path.join(ROOT, "tutel_amd", "csrc", f), "rb").read())
    return h.hexdigest()


def gemm_probe(E_loc, R, M, H, dtype, iters=30):  # noqa: D401
    """the fc1 grouped GEMM (bias + ReLU fused) on random operands at one shape, alternating two weight sets so that the weights
    come from HBM: average launch in us between HIP events around `iters` back-to-back launches"""
    from tutel_amd import ops
    dev = torch.device("cuda", torch.cuda.current_device())
    g = torch.Generator(device=dev).manual_seed(11)
    a = torch.randn([E_loc, R, M], device=dev, generator=g).to(dtype)
    ws = [(torch.randn([E_loc, H, M], device=dev, generator=g) * 0.03).to(dtype) for _ in range(2)]
    b = torch.randn([E_loc, H], device=dev, generator=g).to(dtype)
    for i in range(6):
        ops.expert_gemm(a, ws[i & 1], b, True, act="relu")
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for i in range(iters):
        ops.expert_gemm(a, ws[i & 1], b, True, act="relu")
    e.record()
    torch.cuda.synchronize()
    us = s.elapsed_time(e) * 1e3 / iters
    flops = 2.0 * E_loc * R * M * H
    return {"shape": f"{E_loc} experts x {R} rows, K = {M}, N = {H}", "avg_launch_us": round(us, 2), "tflops": round(flops / us * 1e-6, 1),
            "frac_of_mfma_peak": round(flops / us * 1e-6 / MFMA_PEAK_TFLOPS, 4), "launches_timed": iters, "bound": "mfma"}


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
    """the reference's CPU path at the headline shape, timed on this box's host cores (rank 0 at N = 1, after every GPU pass).
    kind "reference": fast_encode / fast_decode run the REFERENCE'S OWN compiled CPU kernels (oracle/_ref/tutel_custom_kernel.so =
    tutel/custom/custom_kernel.cpp built where it lies; the .so travels to the GPU box, the reference's Python package does not),
    called as its Python calls them (oracle/ref_kernels.py), between the ATen calls that Python makes (softmax, top-k, matmul:
    oracle/moe_oracle.py).  The plain-C port of the same loops is timed beside it (`port`), and is `value` only where _ref is absent."""
    from oracle import moe_oracle as O, ref_kernels as R
    threads = min(physical_cores(), 32)  # ATen GEMMs stop scaling (and the scatter loops are single-threaded) well before that
    old = torch.get_num_threads()
    torch.set_num_threads(threads)

    def median_forward(fwd, budget):
        with torch.no_grad():
            fwd()  # warm-up (page-in, thread pool)
            times, t_all = [], time.time()
            while len(times) < 10 and (time.time() - t_all) < budget:
                t0 = time.time()
                fwd()
                times.append(time.time() - t0)
        return sorted(times)[len(times) // 2], len(times)
    try:
        x, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=torch.float32, seed=0)
        have_ref = R.available()
        if have_ref:
            with torch.no_grad():   # the two must agree before either is quoted
                ya, yb = O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k)[0], R.moe_forward(x, wg, w1, b1, w2, b2, top_k=k)[0]
            assert torch.equal(ya, yb), "oracle/_ref kernels and the port disagree"
            dt_ref, n_ref = median_forward(lambda: R.moe_forward(x, wg, w1, b1, w2, b2, top_k=k), max_seconds / 2)
        dt_port, n_port = median_forward(lambda: O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k), max_seconds / 2)
    finally:
        torch.set_num_threads(old)
    dt, n = (dt_ref, n_ref) if have_ref else (dt_port, n_port)
    note = ("oracle/_ref/tutel_custom_kernel.so -- the reference's tutel/custom/custom_kernel.cpp compiled where it lies (oracle/Makefile), its "
            "invoke_cpu_fp32 scatter / gather kernels called as tutel/impls/fast_dispatch.py calls them (oracle/ref_kernels.py) -- between the "
            "ATen calls the reference's Python makes on its CPU path (softmax, top-k, cumsum, matmul: oracle/moe_oracle.py); the Python package "
            "itself cannot travel to the GPU box (/root/reference does not exist there)") if have_ref else \
           ("oracle/moe_oracle.py moe_forward: the reference's CPU path restated (ATen softmax / matmul + C scatter / gather loops); "
            "oracle/_ref/ is not on this box, so no reference code can be timed here")
    return {"value": round(T / dt, 1), "unit": "tokens/s", "cores": threads, "kind": "reference" if have_ref else "port",
            "kind_note": note, "ms_per_step": round(dt * 1e3, 2), "dtype": "f32", "host_logical_cpus": os.cpu_count(), "host_physical_cores": physical_cores(),
            "sample": f"median of {n} forward passes of the same {T}-token workload (fp32), torch.set_num_threads({threads}); outputs of the "
                      f"reference kernels and of the port compared bit for bit first",
            "port": {"value": round(T / dt_port, 1), "ms_per_step": round(dt_port * 1e3, 2), "kind": "port", "passes": n_port,
                     "note": "oracle/moe_oracle.c loops instead of the reference's compiled kernels, everything else the same"},
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


def free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def launcher_command(n, argv):
    """what `python bench.py --gpus N` execs when no launcher set WORLD_SIZE: the driver's own multi-GPU command line"""
    return [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n), "--master-addr", "127.0.0.1",
            "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)


def maybe_relaunch(args):
    """--gpus N > 1 without a launcher (no WORLD_SIZE in the environment): start N rank processes of this script under
    torch.distributed.run and become that launcher (the bare command used to die on an assertion)."""
    if args.gpus <= 1 or "WORLD_SIZE" in os.environ:
        return
    cmd = launcher_command(args.gpus, sys.argv[1:])
    if os.environ.get("TUTEL_AMD_BENCH_LAUNCH_ECHO") == "1":   # CPU test hook: show the command instead of running it
        print(json.dumps({"would_exec": cmd}), flush=True)
        sys.exit(0)
    share = os.environ.get("TUTEL_AMD_BENCH_SHARE_GPU", "0") == "1"
    have = torch.cuda.device_count() if torch.cuda.is_available() else 0
    if have < args.gpus and not share:
        print(f"bench.py: --gpus {args.gpus} but this box exposes {have} GPU(s) (TUTEL_AMD_BENCH_SHARE_GPU=1 runs the ranks on one "
              f"device as a code-path test, not a measurement)", file=sys.stderr, flush=True)
        sys.exit(2)
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    if share and args.gpus >= 4:
        env.setdefault("GPU_MAX_HW_QUEUES", "2")   # ranks sharing ONE device: stay inside its hardware queue slots (tests/test_ep_ranks_one_gpu.py)
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def parity_of(y, ref):
    """max |y - ref| and whether the two are the same bits; ok = within one bf16 / fp16 ulp of the reference's scale (the paths run the
    same kernels in the same order, so the expectation is bitwise equality -- a row of another batch is off by O(scale))"""
    d = (y.float() - ref.float()).abs()
    scale = float(ref.float().abs().max())
    err = float(d.max()) if d.numel() else 0.0
    fin = bool(torch.isfinite(y.float()).all())
    return {"max_abs_err": err, "scale": scale, "bitwise": bool(torch.equal(y, ref)), "ok": fin and err <= 2 ** -6 * max(scale, 1e-6)}


def single_rank_parity(layer, step, T, M, k, dtype, dev, graphed):
    """N = 1: the parity canary the N > 1 branch has had since round 5 -- before anything is timed, a SECOND random batch goes through the
    timed path (the HIP-graph replay when there is one) and is compared with the CPU oracle routed on the kernel's own scores (softmax is
    not bit-specified across exp implementations; everything after it is): every (token, choice) expert id and bucket slot must be the
    oracle's -- ties in torch.topk's CPU order -- and the output must sit within the dtype's bar of the oracle's fp32-accumulated FFN.
    The oracle is the CHECKER here (test infrastructure, like the cpu_baseline leg); a forward that fails is never timed."""
    from oracle import moe_oracle as O
    from tutel_amd import ops
    g = torch.Generator().manual_seed(20250601)
    x2 = torch.randn([16, T // 16, M], generator=g, dtype=torch.float32).to(dtype)
    keep = getattr(layer, "_keep_routing", False)
    layer._keep_routing, layer.last_logits, layer.last_routing = True, None, None
    replay_equal = None
    try:
        with torch.no_grad():
            y = step(x2.to(dev)).clone()          # the eager forward: its routing arrays and logits are read back (`_keep_routing`)
            cnt = layer.dispatch_count.clone()
            logits = layer.last_logits.clone() if layer.last_logits is not None else layer.gates[0](x2.to(dev).view(-1, M))
            idx, loc = [t.clone() for t in layer.last_routing]
            if graphed is not None:               # the captured forward reads its tokens from the static input buffer: same batch, same bits?
                saved = graphed.static_in.clone()
                graphed.static_in.copy_(x2.to(dev))
                yg = graphed(graphed.static_in).clone()
                replay_equal = bool(torch.equal(yg, y)) and bool(torch.equal(layer.dispatch_count, cnt))
                graphed.static_in.copy_(saved)
            torch.cuda.synchronize()
            scores = ops.gate_topk(logits, k, apply_softmax=True, want_scores=True)[3].cpu()
    finally:
        layer._keep_routing = keep
    ex = layer.experts
    w1, b1 = ex.batched_fc1_w.detach().cpu(), ex.batched_fc1_bias.detach().cpu()
    w2, b2 = ex.batched_fc2_w.detach().cpu(), ex.batched_fc2_bias.detach().cpu()
    cf = float(layer.gates[0].capacity_factor)
    from tutel_amd import _lib as _L
    rule, O.TIE_RULE = O.TIE_RULE, ("lowest" if ops.get_option(_L.OPT_TIE_RULE) == 0 else "aten")   # the checker follows TUTEL_OPT_TIE_RULE
    try:
        with torch.no_grad():
            crit, _ = O.extract_critical(scores, k, cf)
    finally:
        O.TIE_RULE = rule
    with torch.no_grad():
        ok_idx = bool(torch.equal(torch.stack(crit[1]).to(torch.int32), idx.cpu())) and bool(torch.equal(torch.stack(crit[2]), loc.cpu()))
        ok_cnt = bool(torch.equal(cnt.cpu(), crit[5]))
        enc = O.fast_encode(x2.view(-1, M), crit)
        yo = O.fast_decode(O.expert_ffn(enc, w1, b1, w2, b2, accum_fp32=True), crit)
    err = (y.view(-1, M).float().cpu().double() - yo.double()).abs()
    rel = 2 ** -7 if dtype == torch.bfloat16 else 2 ** -10
    ok_y = bool(torch.isfinite(y.float()).all()) and bool((err <= rel * yo.double().abs() + 2e-3).all())
    return {"ok": ok_idx and ok_cnt and ok_y and replay_equal is not False, "assignments_checked": int(idx.numel()), "assignment_exact": ok_idx,
            "dispatch_count_exact": ok_cnt, "max_abs_err": float(err.max()), "output_scale": float(yo.double().abs().max()),
            "bar": f"|err| <= {rel} * |ref| + 2e-3", "hip_graph_replay_bitwise_equal_to_the_checked_eager_forward": replay_equal,
            "through": "eager forward vs the oracle" + ("; the HIP-graph replay (the timed path) vs that eager forward, bit for bit" if graphed is not None else ""),
            "reference": "oracle/moe_oracle.py (extract_critical with torch.topk's CPU tie order, fast_encode, expert_ffn with fp32 accumulation, fast_decode) on the "
                         "kernel's own scores, a second random batch"}


EP_TRANSPORTS = ("rccl", "ipc", "torch")   # `value` = the first that passes its parity canary: RCCL all_to_all (north_star) before the peer stores


def configure_ep(layer, transport, degree):
    """switch the layer's exchange (collective: every rank calls it at the same point).  "torch" = the Python-orchestrated pipeline
    over torch.distributed (impls/overlap.py); "ipc" / "rccl" = the one-call native pipeline over that transport."""
    from tutel_amd.impls import ep_native as EN
    torch.cuda.synchronize()   # nothing of the previous mode may still be running when its segments are unmapped ...
    dist.barrier()             # ... on ANY rank (peers store into them)
    EN.set_transport("ipc" if transport == "ipc" else "rccl")
    EN.ENABLED = transport != "torch"
    EN.forget_workspaces(layer)
    layer.a2a_ffn_overlap_degree = degree


def ep_transport_running(layer, dev):
    """which exchange the layer's last forward really used"""
    from tutel_amd.impls import ep_native as EN
    if not EN.ENABLED:
        return "torch"
    comm = EN.communicator(layer.group, dev) if EN.group_ok(layer.group) else None
    if comm is None:
        return "torch"
    return "ipc" if comm.ipc else ("rccl" if not hasattr(comm, "register") else "hosted")


def agree_min(v, share, dev):
    f = torch.tensor([int(v)], device="cpu" if share else dev, dtype=torch.int32)
    dist.all_reduce(f, op=dist.ReduceOp.MIN)
    return int(f)


def ep_sweep(layer, batches, world, rank, dev, share, steps, warmup, gate_timer):
    """reference outputs over torch.distributed, then every (transport, degree): parity canary, short timed run.
    Returns (list of mode records, reference outputs)."""
    from tutel_amd.impls import ep_native as EN
    from tutel_amd.impls.graph import GraphedForward
    T = batches[0].shape[0] * batches[0].shape[1]
    with torch.no_grad():
        configure_ep(layer, "torch", 1)
        refs = [layer(b).clone() for b in batches]
        torch.cuda.synchronize()
    modes = []
    for transport in EP_TRANSPORTS:
        for degree in (2, 1):
            rec = {"transport": transport, "a2a_ffn_overlap_degree": degree, "available": False, "parity": None, "launch": None,
                   "ms_per_step": None, "value": None, "note": None}
            ok, step = 1, None
            try:
                with torch.no_grad():
                    configure_ep(layer, transport, degree)
                    y0 = layer(batches[0])
                    torch.cuda.synchronize()
                    running = ep_transport_running(layer, dev)
                    if running != transport:
                        rec["note"] = f"not available here (the forward ran over {running})"
                        ok = 0
                    else:
                        rec["available"] = True
                        ys = [y0.clone()] + [layer(b).clone() for b in batches[1:]] + [layer(batches[0]).clone()]   # ... and the first one again
                        torch.cuda.synchronize()
                        EN.ipc_status()
                        ps = [parity_of(y, refs[i % len(refs)]) for i, y in enumerate(ys)]
                        rec["parity"] = {"checked": len(ps), "max_abs_err": max(q["max_abs_err"] for q in ps), "bitwise": all(q["bitwise"] for q in ps),
                                         "ok": all(q["ok"] for q in ps), "reference": "torch.distributed all_to_all_single path, degree 1"}
                        ok = int(rec["parity"]["ok"])
            except Exception as ex:   # noqa: BLE001 -- loud, and agreed below
                rec["note"] = f"{type(ex).__name__}: {str(ex)[:240]}"
                ok = 0
            passed = agree_min(ok, share, dev)
            if rec["available"] and rec["parity"] is not None:
                rec["parity"]["ok_on_every_rank"] = bool(passed)
            if not passed:
                if rec["available"] and rank == 0:
                    print(f"bench.py: exchange '{transport}' degree {degree} FAILED its parity canary / raised ({rec['note'] or rec['parity']}): not timed",
                          file=sys.stderr, flush=True)
                modes.append(rec)
                continue
            tok = 1
            try:
                with torch.no_grad():
                    step, rec["launch"] = layer, "eager"
                    x = batches[0]
                    if transport == "ipc":   # plain kernels + events: the captured forward replays
                        try:
                            g = GraphedForward(layer, batches[0])
                            step, x, rec["launch"] = g, g.static_in, "hip-graph replay"
                        except Exception as ex:   # noqa: BLE001
                            rec["note"] = f"graph capture failed ({type(ex).__name__}: {str(ex)[:120]}); eager"
                    gk = agree_min(rec["launch"] != "eager", share, dev)
                    if not gk and rec["launch"] != "eager":
                        step, x, rec["launch"] = layer, batches[0], "eager"
                    for _ in range(max(warmup, 10)):
                        step(x)
                    el, _, _, _ = run_timed(step, x, steps, world, gate_timer, mode=0, marks=False)
                    EN.ipc_status()
            except Exception as ex:   # noqa: BLE001
                rec["note"] = f"{type(ex).__name__}: {str(ex)[:240]}"
                tok, el = 0, 0.0
            tt = torch.tensor([el if tok else 1e9], device="cpu" if share else dev, dtype=torch.float64)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            if float(tt) < 1e8:
                rec["ms_per_step"] = round(float(tt) / steps * 1e3, 4)
                rec["value"] = round(world * T / (float(tt) / steps), 1)
            step = None
            modes.append(rec)
    return modes, refs


def rank_pipeline_probe(n, M, H, T, k, dtype, dev, iters=60):
    """ONE rank's pipeline of an n-way expert-parallel run of this workload, alone on this GPU, through a world-size-1 IPC communicator
    (the peer is the rank itself: fast_encode and the fc2 epilogue store through the peer table, flags are signalled and awaited as
    between ranks): E_loc = E / n local experts x n * capacity rows -- a 1-rank layer with E_loc experts over the same T tokens has
    exactly those GEMM, encode and decode shapes.  Returns ms per forward for degree 1 and 2 (eager wall over `iters` back to back)."""
    from tutel import moe
    from tutel_amd.impls import ep_native as EN
    E_loc = 64 // n
    torch.set_default_dtype(dtype)
    try:
        layer = moe.moe_layer(gate_type={"type": "top", "k": k}, model_dim=M,
                              experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)}).to(dev).eval()
    finally:
        torch.set_default_dtype(torch.float32)
    x = torch.randn([T, M], device=dev).to(dtype)
    out = {}
    with torch.no_grad():
        for degree in (1, 2):
            EN.forget_workspaces(layer)
            for _ in range(10):
                layer(x, a2a_ffn_overlap_degree=degree)
            torch.cuda.synchronize()
            comm = EN.communicator(layer.group, dev)
            assert comm is not None and comm.ipc and comm.world == 1
            t0 = time.perf_counter()
            for _ in range(iters):
                layer(x, a2a_ffn_overlap_degree=degree)
            torch.cuda.synchronize()
            out[degree] = (time.perf_counter() - t0) / iters * 1e3
    del layer
    return out


XGMI_LINK_GBS = 153.0   # per link and direction (MI355X_MICROARCH.md: 7 links x ~153 GB/s per GPU, one link per peer)


def modelled_scaling(M, H, T, E, k, C, dtype, dev):
    """SURVEY 8(d)'s fallback for the points a one-GPU box cannot measure: per-rank compute MEASURED here at the per-rank shapes + a link
    time MODEL.  Every number in here is labelled modelled; none of it is `value`."""
    from tutel_amd.impls import ep_native as EN
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", str(free_port()))
    made_pg = False
    if not dist.is_initialized():
        dist.init_process_group("gloo", rank=0, world_size=1)
        made_pg = True
    old = (EN.HOSTED, EN.TRANSPORT, EN._FORCE_COMM)
    EN.set_transport("ipc", hosted=False)
    EN._FORCE_COMM = True
    points = {}
    try:
        for n in (2, 4, 8):
            t = rank_pipeline_probe(n, M, H, T, k, dtype, dev)
            per_peer = E * C * M * 2 / n                       # bytes one rank sends to ONE peer per direction (its own link)
            link_ms = per_peer / (XGMI_LINK_GBS * 1e9) * 1e3   # all n - 1 peers in parallel, one link each
            serial = t[1] + 2 * link_ms                        # degree 1: nothing hidden
            half = t[2] + link_ms                              # degree 2: one of the two stages' transfers hidden behind the other's GEMMs, each way
            best = min(serial, half)
            points[str(n)] = {"modelled": True, "rank_pipeline_ms_measured": {"degree1": round(t[1], 4), "degree2": round(t[2], 4)},
                              "per_peer_MB_per_direction": round(per_peer / 1e6, 2), "link_ms_per_direction_model": round(link_ms, 4),
                              "ms_per_step_model": {"degree1_nothing_hidden": round(serial, 4), "degree2_half_hidden": round(half, 4)},
                              "tokens_per_s_model": round(n * T / (best * 1e-3), 1)}
    finally:
        EN.destroy_all()
        EN.HOSTED, EN.TRANSPORT, EN._FORCE_COMM = old
        if made_pg:
            dist.destroy_process_group()
    return {"modelled": True, "points": points,
            "model": "tokens/s(N) = N * T / (one rank's pipeline, MEASURED alone on this GPU through a world-size-1 IPC communicator at E_loc = 64/N experts x "
                     f"N * {C} rows, + link time), link time per direction = E*C*M*2/N bytes to each peer over its own xGMI link at {XGMI_LINK_GBS} GB/s "
                     "(MODEL; no link was exercised), degree 1: both directions exposed, degree 2: half of them hidden",
            "note": "NOT a measurement of a multi-GPU run: the compute legs are measured on one GPU, the exchange is a formula"}


def round5_features(layer, x, fwd_kw, gate_timer):
    """the changes of rounds 5 and 6 to the headline path, each switched in turn: ms per step of the HIP-graph replay of the same forward
    (median of 3 x 20 steps).  bench.py runs it in a child process (`--round5_features`): every capture brings a stream of its own."""
    from tutel_amd import _lib, ops
    from tutel_amd.impls import moe_layer as _ML
    from tutel_amd.impls.graph import GraphedForward as _GF
    feat = {}
    names = ["all on (as timed)", "location kernel instead of the in-GEMM scan",
             "fused location off + vector slot-map lookups in front of the weight DMA",
             "gate projection by F.linear (hipBLASLt) instead of the split-K kernel",
             "plain (write-back) output stores in the expert GEMMs instead of write-through",
             "round 6: exact ties by lowest expert index instead of the reference's CPU torch.topk order (TUTEL_OPT_TIE_RULE = 0; NOT the reference's assignment)",
             "round 6: fc1 -> activation -> fc2 in one persistent launch (TUTEL_OPT_FFN_FUSED = 1; opt-in, same bits)"]
    # (TUTEL_OPT_FUSED_LOCATION, TUTEL_OPT_GEMM_GATHER, native gate, TUTEL_OPT_GEMM_STORE, TUTEL_OPT_TIE_RULE, TUTEL_OPT_FFN_FUSED)
    settings = [(-1, -1, True, -1, -1, -1), (0, -1, True, -1, -1, -1), (0, 0, True, -1, -1, -1), (-1, -1, False, -1, -1, -1), (-1, -1, True, 0, -1, -1),
                (-1, -1, True, -1, 0, -1), (-1, -1, True, -1, -1, 1)]
    graphs = []
    try:
        # one capture per setting (the kernel choice is made at capture time, each graph keeps its own workspace), then the
        # graphs are replayed INTERLEAVED, three rounds: clocks and neighbours drift by more than the differences being measured
        for fl, ga, ng, st, tr, ff in settings:
            ops.set_option(_lib.OPT_FUSED_LOCATION, fl)
            ops.set_option(_lib.OPT_GEMM_GATHER, ga)
            ops.set_option(_lib.OPT_GEMM_STORE, st)
            ops.set_option(_lib.OPT_TIE_RULE, tr)
            ops.set_option(_lib.OPT_FFN_FUSED, ff)
            _ML._NATIVE_GATE = ng
            layer.__dict__.pop("_ep_workspaces", None)
            with torch.no_grad():
                graphs.append(_GF(layer, x, **fwd_kw))
        with torch.no_grad():
            for g in graphs:
                for _ in range(100):
                    g(g.static_in)
            ts = [[] for _ in graphs]
            for _ in range(3):
                for i, g in enumerate(graphs):
                    for _ in range(20):
                        g(g.static_in)
                    ts[i].append(run_timed(g, g.static_in, 20, 1, gate_timer, mode=0, marks=False)[0] / 20 * 1e3)
        for name, t in zip(names, ts):
            feat[name] = round(sorted(t)[1], 4)
    except Exception as ex:   # noqa: BLE001 -- an extra must never cost the line
        feat["error"] = f"{type(ex).__name__}: {str(ex)[:200]}"
    finally:
        _ML._NATIVE_GATE = True
        for key in (_lib.OPT_GEMM_GATHER, _lib.OPT_FUSED_LOCATION, _lib.OPT_GEMM_STORE, _lib.OPT_TIE_RULE, _lib.OPT_FFN_FUSED):
            ops.set_option(key, -1)
        layer.__dict__.pop("_ep_workspaces", None)
    return feat


def tie_rule(dname):
    """north_star asks for bit-exact token-to-expert assignment against the reference's CPU path.  Among EXACTLY equal scores that path
    returns what ATen's CPU torch.topk leaves (nth_element / partial_sort over (value, index) pairs); since round 6 the top-k kernels
    replay exactly that on the rows that tie (csrc/topk_ties.h, TUTEL_OPT_TIE_RULE).  What is left at this very configuration is
    measured against the reference-written fixture and committed (tests/test_layer_gpu.py::test_headline_low_precision_gate_assignment_vs_reference)."""
    from tutel_amd import _lib, ops
    mode = ops.get_option(_lib.OPT_TIE_RULE)
    if mode == 0:
        return "TUTEL_OPT_TIE_RULE = 0: lowest expert index on exact ties (rounds 1-5); differs from the reference's CPU torch.topk on tied rows"
    path = os.path.join(ROOT, "profiles", f"r06_headline_gate_assignment_{'bfloat16' if dname == 'bf16' else 'float16'}.json")
    try:
        d = json.load(open(path))
        return (f"exact ties in the order of the reference's CPU torch.topk; at this configuration with a {d['dtype']} gate "
                f"{d['differing_tokens_with_the_references_logits_and_scores']} tokens whose logits and scores carry the reference's bits route "
                f"differently ({d['tied_rows_with_the_references_bits']} of them tie at the k / k+1 boundary); {d['differing_assignments']} of "
                f"{d['assignments']} (token, k) assignments differ in all, each on a token where a logit or a score rounds the other way in its "
                f"last bit ({os.path.relpath(path, ROOT)})")
    except Exception:   # noqa: BLE001
        return "exact ties in the order of the reference's CPU torch.topk (csrc/topk_ties.h)"


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
    ap.add_argument("--graph", action="store_true", help="(default when capacity_factor > 0) replay the forward from a captured HIP graph")
    ap.add_argument("--round5_features", action="store_true", help="(internal) print the round-5 feature A/B of the headline forward and exit")
    ap.add_argument("--eager", action="store_true", help="time the Python-enqueued forward instead of the HIP-graph replay")
    args = ap.parse_args()
    maybe_relaunch(args)

    world = int(os.environ.get("WORLD_SIZE", 1))
    rank = int(os.environ.get("RANK", 0))
    local_rank = int(os.environ.get("LOCAL_RANK", 0))
    assert world == args.gpus or world == 1 and args.gpus == 1, f"--gpus {args.gpus} but WORLD_SIZE={world}"
    # Test hook (not a measurement mode): TUTEL_AMD_BENCH_SHARE_GPU=1 runs every rank on cuda:0 with a gloo
    # rendezvous and the IPC transport between the rank processes, so the N > 1 code path of this script can be
    # exercised on a single-GPU box.  The driver's multi-GPU runs use one GPU per rank ("nccl" rendezvous).
    share = os.environ.get("TUTEL_AMD_BENCH_SHARE_GPU", "0") == "1"
    if share:
        local_rank = 0
        os.environ.setdefault("TUTEL_AMD_EP_TRANSPORT", "ipc")   # processes that share a device can still map each other's segments
    if world > 1:   # before the first HIP call of the process: the ROCm runtime reads it when it initialises (dmabuf IPC handles)
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    local_rank %= max(1, torch.cuda.device_count())   # a launcher that hands every rank ONE visible device numbers it 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
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
    eager_step = step
    launch, graph_note, graphed = "eager", None, None
    if args.round5_features:   # child process of the N = 1 run (see round5_features): nothing but this A/B, on a process of its own
        print("ROUND5_FEATURES " + json.dumps(round5_features(layer, x, fwd_kw, gate_timer)), flush=True)
        return
    # ---- N > 1: reference outputs, every exchange x degree through its parity canary and a short timed run, then the choice ----
    ep_modes, parity, chosen, batches, refs, fell_back = None, None, None, None, None, []
    if world > 1:
        ep_native.IPC_TIMEOUT_MS = min(ep_native.IPC_TIMEOUT_MS, 30000)   # a peer that never arrives costs this script 30 s, not the 2-minute watchdog
        g = torch.Generator().manual_seed(4321 + rank)
        batches = [x] + [torch.randn([16, T // 16, M], generator=g, dtype=torch.float32).to(dtype).to(dev) for _ in range(2)]
        ep_modes, refs = ep_sweep(layer, batches, world, rank, dev, share, min(args.steps, 50), args.warmup, gate_timer)
        by = {(m["transport"], m["a2a_ffn_overlap_degree"]): m for m in ep_modes}
        for t in EP_TRANSPORTS:
            m = by.get((t, overlap)) or by[(t, 2)]
            if m["value"] is not None:
                chosen = t
                break
            if m["available"]:
                fell_back.append(t)
        if chosen is None:
            print("bench.py: NO exchange passed its parity canary -- nothing to time", file=sys.stderr, flush=True)
            if rank == 0:
                print(json.dumps({"metric": "MoE-layer fwd tokens/sec, 4096 tok x H=2048 x E=64 top-2", "value": None, "n_gpus": world,
                                  "error": "no exchange passed its parity canary", "ep_modes": ep_modes}), flush=True)
            sys.exit(3)
        if fell_back and rank == 0:
            print(f"bench.py: FELL BACK from {fell_back} to '{chosen}' (parity canary / availability, see ep_modes)", file=sys.stderr, flush=True)
        configure_ep(layer, chosen, overlap)
    # N > 1: the graph is replayed when the exchange is the IPC transport (kernels + events only); GraphedForward refuses to capture
    # RCCL collectives (replaying them hung after ~200 replays in the 1-rank-communicator probe, profiles/r03_ep_streams.txt) and
    # the forward is then timed eager -- every rank takes the same branch (the transport is agreed at communicator creation).
    # The replay is the timed mode (as in round 3) and the eager forward is timed beside it (`value_eager`, `launch_modes.other`): in
    # steady state the two cost the same (0.256-0.259 ms per forward on the round-4 boxes), but a 20-step region entered from a
    # synchronize costs the eager loop its first forward at host pace (0.43-0.44 ms: +8 us on the mean of 20), the replay +1 us.
    want_graph = (args.graph or not args.eager) and args.capacity_factor > 0 and (world == 1 or chosen == "ipc")
    also_graph = args.eager and args.capacity_factor > 0 and world == 1
    if want_graph:
        # same kernels (and, N > 1, the same RCCL collectives on the caller's stream), enqueued by ONE hipGraphLaunch per step:
        # the host cost of a forward drops from ~0.09 (N = 1) / ~0.16 ms (N > 1, degree 2) to one launch, so the step is
        # GPU-paced from the first one after the synchronize.  All ranks agree on the outcome (a rank that fell back alone would
        # still match its peers' collectives -- the graph replays the same calls -- but the line should say what ran).
        from tutel_amd.impls.graph import GraphedForward
        ok = 1
        try:
            with torch.no_grad():
                graphed = GraphedForward(layer, x, **fwd_kw)
        except Exception as ex:   # noqa: BLE001 -- loud, not silent
            ok, graph_note = 0, f"HIP-graph capture not used ({type(ex).__name__}: {str(ex)[:200]}); timed eager instead"
            print("bench.py: " + graph_note, file=sys.stderr, flush=True)
        if world > 1:
            f = torch.tensor([ok], device="cpu" if share else dev, dtype=torch.int32)
            dist.all_reduce(f, op=dist.ReduceOp.MIN)
            if int(f) == 0 and ok:
                ok, graph_note = 0, "HIP-graph capture failed on another rank; timed eager instead"
        if ok:
            x = graphed.static_in   # the tokens live where the captured forward reads them: no per-step copy
            step, launch = graphed, "hip-graph replay"
    elif args.capacity_factor <= 0:
        graph_note = "dropless routing reads the capacity back to the host every step: not capturable, eager"

    if world > 1:
        # the parity canary once more, THROUGH THE TIMED PATH (the graph replay when there is one): three different batches and the
        # first again, every rank against the torch.distributed reference
        def canary(fn):
            with torch.no_grad():
                ys = [fn(b).clone() for b in batches + [batches[0]]]
            torch.cuda.synchronize()
            ep_native.ipc_status()
            ps = [parity_of(yy, refs[i % len(refs)]) for i, yy in enumerate(ys)]
            return {"checked": len(ps), "max_abs_err": max(q["max_abs_err"] for q in ps), "bitwise": all(q["bitwise"] for q in ps),
                    "ok": bool(agree_min(all(q["ok"] for q in ps), share, dev))}
        parity = canary(step)
        if not parity["ok"] and launch != "eager":
            graph_note = "the HIP-graph replay FAILED the parity canary; timed eager instead"
            print("bench.py: " + graph_note, file=sys.stderr, flush=True)
            step, launch, x = eager_step, "eager", batches[0]
            parity = canary(step)
        parity.update(transport=chosen, launch=launch, fell_back_from=fell_back,
                      reference="the same forward over torch.distributed (all_to_all_single on the process group, impls/overlap.py), degree 1")
        if not parity["ok"]:
            print("bench.py: the timed path FAILED its parity canary -- refusing to report a throughput for wrong tokens", file=sys.stderr, flush=True)
            if rank == 0:
                print(json.dumps({"metric": "MoE-layer fwd tokens/sec, 4096 tok x H=2048 x E=64 top-2", "value": None, "n_gpus": world,
                                  "error": "timed path failed its parity canary", "parity": parity, "ep_modes": ep_modes}), flush=True)
            sys.exit(3)
        if graphed is not None and step is graphed:
            graphed.static_in.copy_(batches[0])
            x = graphed.static_in
    if world == 1 and args.capacity_factor > 0 and not args.megablocks_size:
        parity = single_rank_parity(layer, eager_step, T, M, k, dtype, dev, graphed if step is graphed else None)
        if not parity["ok"]:
            print("bench.py: the timed path FAILED its parity canary -- refusing to report a throughput for wrong tokens", file=sys.stderr, flush=True)
            print(json.dumps({"metric": "MoE-layer fwd tokens/sec, 4096 tok x H=2048 x E=64 top-2", "value": None, "n_gpus": world,
                              "error": "timed path failed its parity canary", "parity": parity}), flush=True)
            sys.exit(3)
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
        eager = eager_step
        other, other_launch = None, None
        if launch != "eager":      # the same K steps, eager, bracketed the same way: reported beside the value
            for _ in range(max(3, args.warmup, min(args.settle, 50))):   # (the workspace of the eager stream has not run since the capture)
                eager(x)
            other, _, _, _ = run_timed(eager, x, args.steps, world, gate_timer, mode=0, marks=False)
            other_launch = "eager"
        elif also_graph:           # ... or, when eager is the timed mode, the HIP-graph replay of the same forward beside it
            try:
                from tutel_amd.impls.graph import GraphedForward
                graphed = GraphedForward(layer, x, **fwd_kw)
                for _ in range(max(3, args.warmup, min(args.settle, 50))):
                    graphed(graphed.static_in)
                other, _, _, _ = run_timed(graphed, graphed.static_in, args.steps, world, gate_timer, mode=0, marks=False)
                other_launch = "hip-graph replay"
            except Exception as ex:   # noqa: BLE001
                graph_note = f"HIP-graph replay not timed ({type(ex).__name__}: {str(ex)[:160]})"
        _, per_step, _, _ = run_timed(step, x, nb, world, gate_timer, mode=0, marks=True)     # one mark per step: min / median
        run_timed(eager, x, 3, 1, gate_timer, mode=2, marks=False)                            # (fills the event pool)
        _, _, gemms, _ = run_timed(eager, x, nb, world, gate_timer, mode=2, marks=False)      # events around fc1 / fc2: roofline
        run_timed(eager, x, 3, 1, gate_timer, mode=1, marks=False)
        gate_timer.events.clear()
        _, _, stages, _ = run_timed(eager, x, nb, world, gate_timer, mode=1, marks=False)     # events around every launch: stages
    if world > 1:
        tt = torch.tensor([elapsed, other or 0.0], device="cpu" if share else dev, dtype=torch.float64)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed, other = float(tt[0]), (float(tt[1]) if other is not None else None)
    assert torch.isfinite(y.float()).all()
    if world > 1:
        ep_native.ipc_status()   # an exchange that gave up inside the timed region (its outputs are NaN) must not become a number

    C = int(layer.protected_shape[1]) // world  # capacity the layer actually used (dropless: max expert load)
    R = world * C                               # rows per local expert
    stage_us = {name: round(tot / nb, 2) for name, (tot, cnt) in stages.items() if cnt}
    launches = {name: cnt for name, (tot, cnt) in stages.items() if cnt}
    if gate_timer.events:   # the projection ran outside the native call (F.linear -> hipBLASLt): fp32 gate, TUTEL_AMD_NATIVE_GATE=0, ...
        stage_us["gate_projection(hipBLASLt)"] = round(gate_timer.avg_us(), 2)
    # (otherwise stage_us["gate_projection"] is the split-K MFMA kernel inside the call, csrc/gate_proj.hip)
    fc1_tot, fc1_n = gemms["expert_fc1"]
    fc2_tot, fc2_n = gemms["expert_fc2"]
    fc1_us, fc2_us = fc1_tot / max(1, fc1_n), fc2_tot / max(1, fc2_n)   # per LAUNCH (one pipeline stage when N > 1)
    rows_per_launch, experts_per_launch = R, E_loc
    if world > 1 and overlap > 1:
        pl = ep_native.plan(E, world, C, overlap)
        rows_per_launch, experts_per_launch = pl["gemm_rows"], pl["experts_per_stage"]
    gemm_bytes = (experts_per_launch * H * M + experts_per_launch * rows_per_launch * (M + H)) * es
    gemm_flops = 2 * experts_per_launch * rows_per_launch * M * H
    # counter traffic and the profiler's own average of the dominant kernel are NOT measured in this run: they are read from
    # profiles/traffic.json, which tools/profile_r03.sh writes together with the sha256 of the kernel source they were measured
    # on -- a different source means a different kernel, and then both are reported as null with the reason
    traffic, rocprof_us, traffic_note, dec_rocprof_us = None, None, None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    headline = world == 1 and (T, M, H, E, k) == (4096, 2048, 2048, 64, 2) and args.capacity_factor == 1.0 and dtype == torch.bfloat16
    if not headline:
        traffic_note = "profiles/traffic.json holds the headline configuration only"
    elif not os.path.exists(tpath):
        traffic_note = "profiles/traffic.json is missing"
    else:
        tj = json.load(open(tpath))
        have = source_sha()
        if tj.get("expert_gemm_hip_sha256") != have:
            traffic_note = (f"stale: profiles/traffic.json was measured on csrc/expert_gemm.hip sha256 {str(tj.get('expert_gemm_hip_sha256'))[:12]}, "
                            f"the library was built from {have[:12]} -- re-run tools/profile_r05.sh")
        else:
            traffic = tj.get("expert_gemm_fc1_hbm_bytes_per_launch")
            rocprof_us = tj.get("expert_gemm_fc1_avg_us_rocprofv3")
            dec_rocprof_us = tj.get("fast_decode_avg_us_rocprofv3")
            traffic_note = f"{tj.get('source')}; kernel {tj.get('kernel')}; measured at git {tj.get('git_head')}"

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
        fused_loc = "location" not in stage_us
        roofline = {"bound": "hbm", "kernel": f"expert_gemm_big_kernel<{dname},k-major,relu,NI=4,NS=3,BUF,BM=128{',FL' if fused_loc else ''}> (fc1 grouped GEMM: 128 x 256 tile, "
                                              "three-slot LDS-DMA ring" + ("; fast_encode's row gather AND the location step run inside this launch (round 5: no "
                                                                           "compute_location kernel between top-k and fc1; its ~3 us show up here, the 7.3 us "
                                                                           "kernel is gone)" if fused_loc else "") + ")",
                    "achieved": round(gbs, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(gbs / HBM_PEAK_GBS, 4),
                    "frac_of_achievable": round(gbs / HBM_ACHIEVABLE_GBS, 4), "achievable_GBs": HBM_ACHIEVABLE_GBS, "traffic": traffic,
                    "traffic_over_algorithmic": round(traffic / gemm_bytes, 4) if traffic else None, "traffic_note": traffic_note,
                    "avg_launch_us_rocprofv3": rocprof_us,
                    "frac_rocprof": round(gemm_bytes / rocprof_us * 1e-3 / HBM_PEAK_GBS, 4) if rocprof_us else None,
                    "algorithmic_bytes_per_launch": gemm_bytes, "avg_launch_us": round(fc1_us, 2), "launches_timed": fc1_n,
                    "fc2_gemm": fc2_obj, "mfma_tflops_fc1": round(gemm_flops / fc1_us * 1e-6, 1)}

    # second kernel of interest: fast_decode (HBM-bound permutation).  Algorithmic bytes: the kept bucket rows it
    # gathers + the token rows it writes; kept rows from the layer's own dispatch counts.
    n_kept = int(torch.clamp(layer.dispatch_count.to(torch.int64), max=C).sum()) if C > 0 else 0
    dec_bytes = (n_kept + T) * int(layer.protected_shape[-1]) * es
    dec_tot, dec_n = stages.get("fast_decode", (0.0, 0))
    dec_us = dec_tot / max(1, dec_n)
    decode_obj = {"bound": "hbm", "kernel": f"decode_kernel<{dname}, k={k}> (fast_decode)", "algorithmic_bytes_per_launch": dec_bytes,
                  "kept_rows": n_kept, "avg_launch_us": round(dec_us, 2), "launches_timed": dec_n,
                  "achieved": round(dec_bytes / max(dec_us, 1e-9) * 1e-3, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                  "frac": round(dec_bytes / max(dec_us, 1e-9) * 1e-3 / HBM_PEAK_GBS, 4),
                  "frac_of_achievable": round(dec_bytes / max(dec_us, 1e-9) * 1e-3 / HBM_ACHIEVABLE_GBS, 4),
                  "avg_launch_us_rocprofv3": dec_rocprof_us,
                  "frac_rocprof": round(dec_bytes / dec_rocprof_us * 1e-3 / HBM_PEAK_GBS, 4) if dec_rocprof_us else None,
                  "note": "avg_launch_us from timing pass 4 (events around every launch: each record drains the queue, +1-2 us per launch); "
                          "avg_launch_us_rocprofv3 from profiles/traffic.json (kernel trace of the same command, stamped like the fc1 figures)"}
    if rank == 0:
        ms = elapsed / args.steps * 1e3
        value = world * T / (elapsed / args.steps)
        # algorithmic bytes of the whole forward on one rank: gate reads x; encode (T + E*C) rows; FFN weights + rows in/out;
        # decode (n_kept + T) rows; all-to-all bytes are link traffic, not HBM, and are listed separately
        # SURVEY 8(d)'s definition: gate T*M; encode (T + E*C)*M; expert FFN weights 2*E_loc*H*M + rows in / out
        # E_loc*R*(M + M) with the [R, H] intermediate kept ON CHIP; decode (n_kept + T)*M -- 1 257 MB at the headline shape.  (The
        # two-launch FFN really writes and re-reads `hid`, 2 x 33.5 MB more, mostly through the Infinity Cache: `bytes_moved_two_launch_ffn`.)
        layer_bytes = (T * M + (T + E * C) * M + 2 * E_loc * H * M + 2 * E_loc * R * M + (min(k * T, E * C) + T) * M) * es
        moved_bytes = layer_bytes + 2 * E_loc * R * H * es
        srt = sorted(per_step)
        out = {
            "metric": "MoE-layer fwd tokens/sec, 4096 tok x H=2048 x E=64 top-2",
            "value": round(value, 1), "unit": "tokens/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "settle": args.settle, "launch": launch,
            "warmup_note": f"the driver's --warmup {args.warmup} untimed steps run right before the timed region; BEFORE them come {args.settle} more untimed `--settle` "
                           "passes of this script's own (allocator, weight pre-layout, graph capture, clocks) and the parity canary: nothing of either is inside the timed K steps",
            "value_eager": round(value, 1) if launch == "eager" else (round(world * T / (other / args.steps), 1) if other is not None else None),
            "ms_per_step": round(ms, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": dname, "data": "synthetic" if not share else "synthetic; TEST HOOK: all ranks share one GPU (IPC transport between the rank processes) -- not a measurement",
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
                       "exchange": {"ipc": "IPC transport: peer stores over xGMI from fast_encode and the fc2 epilogue, flag kernels, epoch canaries, no collective (tutel_amd_ep_forward)",
                                    "rccl": "ncclAllToAll on the library's RCCL communicator (tutel_amd_ep_forward)",
                                    "torch": "torch.distributed all_to_all_single on a communication stream, GEMMs on the caller's (impls/overlap.py)",
                                    None: "none (single rank)"}[chosen],
                       "tie_rule": tie_rule(dname)},
            "roofline": roofline,
            "stages": {"avg_us_per_step": stage_us, "launches_timed": launches,
                       "sum_us": round(sum(stage_us.values()), 2),
                       "steps": nb,
                       "note": "pass 4: HIP events around EVERY launch (tutel_amd_stage_timing(1)); the sum exceeds a bare step because "
                               "every event record drains the queue"},
            "decode": decode_obj,
            "launch_modes": {"timed": launch, "note": graph_note,
                             "other": (None if other is None else {"launch": other_launch, "ms_per_step": round(other / args.steps * 1e3, 4),
                                                                   "value": round(world * T / (other / args.steps), 1)})},
            "layer_roofline": {"algorithmic_bytes_per_step": layer_bytes, "bytes_moved_two_launch_ffn": moved_bytes,
                               "definition": "SURVEY 8(d): gate + encode + FFN (weights + rows in / out, hidden on chip) + decode",
                               "achieved_GBs": round(layer_bytes / (ms * 1e-3) * 1e-9, 1),
                               "frac_of_hbm_peak": round(layer_bytes / (ms * 1e-3) * 1e-9 / HBM_PEAK_GBS, 4),
                               "frac_of_hbm_achievable": round(layer_bytes / (ms * 1e-3) * 1e-9 / HBM_ACHIEVABLE_GBS, 4)},
        }
        if world == 1 and parity is not None:
            out["parity"] = parity
        if world > 1:
            ok_modes = [m for m in ep_modes if m["value"] is not None]
            bm = max(ok_modes, key=lambda m: m["value"]) if ok_modes else None
            out["parity"] = parity
            out["ep_modes"] = ep_modes
            out["best"] = None if bm is None else {"transport": bm["transport"], "a2a_ffn_overlap_degree": bm["a2a_ffn_overlap_degree"],
                                                   "value": bm["value"], "ms_per_step": bm["ms_per_step"], "launch": bm["launch"],
                                                   "note": f"fastest mode that passed its parity canary ({min(args.steps, 50)} timed steps each); `value` is degree "
                                                           f"{overlap} on '{chosen}' (north_star's configuration), timed over {args.steps} steps"}
        if world == 1 and not args.no_extra and headline:
            # BASELINE configs[2] (same shape, dropless + megablocks): a short secondary measurement, recorded next to the headline one
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
        if world == 1 and not args.no_extra and headline:
            # third kernel of interest: the grouped GEMM at the per-rank shapes of the 8-GPU point of this metric (E_loc = 8, rows =
            # 8 ranks x capacity 128): one pipeline stage of a2a_ffn_overlap_degree 2 (4 experts) and the whole rank (8 experts)
            out.setdefault("extra", {})["ep8_rank_gemms"] = {
                "stage_gemm": gemm_probe(4, 1024, M, H, dtype), "full_rank_gemm": gemm_probe(8, 1024, M, H, dtype),
                "note": "expert_gemm (fc1: bias + ReLU fused) launched alone on one GPU at the shapes an 8-way expert-parallel rank runs; MFMA-bound"}
            # the same for the expert problems of BASELINE configs[3] / [4] (M = H = 4096; configs[4] is fp16, 16 local experts)
            out["extra"]["rank_gemms_configs3_4"] = {
                "configs3_bf16_8x1024x4096x4096": gemm_probe(8, 1024, 4096, 4096, torch.bfloat16, iters=12),
                "configs4_fp16_16x1024x4096x4096": gemm_probe(16, 1024, 4096, 4096, torch.float16, iters=8),
                "note": "per-rank fc1 grouped GEMM of the 8-GPU configurations (E_loc experts x W*C = 1024 rows, K = N = 4096), alone on one GPU"}
            # SURVEY 8(d)'s secondary reading of configs[1]: batch 16 x 4096 tokens = 65 536 tokens through the same layer (2048 rows per expert: MFMA-bound)
            x65 = torch.randn([16, 4096, M], device=dev).to(dtype)
            with torch.no_grad():
                for _ in range(3):
                    layer(x65)
                el65, _, _, _ = run_timed(layer, x65, 10, 1, gate_timer, mode=0, marks=False)
            fl65 = 4.0 * E * int(layer.protected_shape[1]) * M * H
            out["extra"]["tokens_65536"] = {"workload": "configs[1] read as batch 16 x 4096 tokens (SURVEY 8d secondary point): 65 536 tokens, capacity "
                                                        f"{int(layer.protected_shape[1])} rows per expert", "value": round(65536 / (el65 / 10), 1), "unit": "tokens/s",
                                            "ms_per_step": round(el65 / 10 * 1e3, 4), "steps": 10, "launch": "eager",
                                            "expert_tflops": round(fl65 / (el65 / 10) * 1e-12, 1), "frac_of_mfma_peak": round(fl65 / (el65 / 10) * 1e-12 / MFMA_PEAK_TFLOPS, 4)}
            del x65
            try:
                out["extra"]["modelled_scaling"] = modelled_scaling(M, H, T, E, k, C, dtype, dev)
            except Exception as ex:   # noqa: BLE001 -- an extra must never cost the line
                out["extra"]["modelled_scaling"] = {"modelled": True, "error": f"{type(ex).__name__}: {str(ex)[:200]}"}
            # the feature A/B runs in a process of its own: it captures four graphs (four more streams), and a process that has created
            # many streams neither keeps the expert-parallel pipeline's side streams on hardware queues of their own (the probe above
            # measured 0.72 ms instead of 0.22 per degree-2 forward after it) nor replays cleanly itself (0.254 instead of 0.243)
            try:
                import subprocess
                r = subprocess.run([sys.executable, os.path.abspath(__file__), "--round5_features", "--settle", "0", "--no_extra", "--no_cpu_baseline"],
                                   capture_output=True, text=True, timeout=300)
                ln = [l for l in r.stdout.splitlines() if l.startswith("ROUND5_FEATURES ")]
                out["extra"]["feature_ab_ms_per_step"] = json.loads(ln[-1][len("ROUND5_FEATURES "):]) if ln else {"error": (r.stderr or r.stdout)[-300:]}
            except Exception as ex:   # noqa: BLE001
                out["extra"]["feature_ab_ms_per_step"] = {"error": f"{type(ex).__name__}: {str(ex)[:200]}"}
        if world == 1 and not args.no_cpu_baseline:
            out["cpu_baseline"] = cpu_baseline(T, M, H, E, k)
        print(json.dumps(out), flush=True)
    if world > 1:
        ep_native.destroy_all()
        dist.destroy_process_group()
    elif dist.is_initialized():
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
