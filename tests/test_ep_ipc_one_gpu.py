"""The IPC transport of the expert-parallel pipeline (csrc/ep.hip, round 4) with 2 and 4 ranks ON ONE GPU.

The rank processes all map cuda:0 and each other's exchange segments (hipIpcOpenMemHandle works between processes on one
device), so what runs here is the device-side protocol of a multi-GPU node, not an emulation of it: fast_encode stores its
bucket rows into the owner rank's receive array, the second expert GEMM stores its output rows into the source rank's
return array, one flag word per (direction, stage, peer) is written by the signal kernel and polled by the wait kernel, and
no collective is enqueued (the reference: ncclSend / ncclRecv per peer and chunk, custom_kernel.cpp:520-654, driven from
overlap.py:8-67).  Checked against the oracle's W-rank simulation, bit for bit against the host-staged exchange, against
the reference's own multi-rank outputs (tests/golden/ep_*.npz), with unequal / empty ranks, replayed 2000 times from a
HIP graph, and with a peer that never arrives (bounded wait, loud error)."""
import os

import pytest
import torch

from test_ep_ranks_one_gpu import (_ep_fixtures, _fixture_worker, _run_ranks, _set_transport, _sweep_worker, _worker_q_last)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,degree,E_loc", [(2, 1, 2), (2, 2, 4), (2, 2, 3), (2, 4, 4), (4, 2, 2), (4, 1, 1)])
def test_ipc_transport_ranks_sharing_one_gpu(world, degree, E_loc):
    """expert-sliced and capacity-chunked stages, degrees 1 / 2 / 4: vs the oracle, three calls bit-identical, and bit-identical
    to the host-staged exchange of the same forward"""
    res = _run_ranks(_worker_q_last, world, (degree, E_loc, None, "ipc"), timeout=300)
    for rank, ok, info, plans in res:
        if degree > 1:
            assert plans == [E_loc % degree == 0]


def test_ipc_transport_rank_shape_with_256_row_tiles():
    """8 local experts x 1024 rows, M = H = 2048 (the per-rank problem of an 8-way run at the headline dims, reproduced with two
    ranks): the stage GEMMs run the 256-row-tile kernels, whose epilogue goes through LDS -- that epilogue's peer stores"""
    world, degree, E_loc = 2, 2, 8
    _run_ranks(_worker_q_last, world, (degree, E_loc, (4096, 2048, 2048, 2), "ipc"), timeout=900)


@pytest.mark.parametrize("path", [pytest.param(p, id=os.path.basename(p)[3:-4]) for p in _ep_fixtures()])
def test_ipc_transport_vs_reference_multi_rank_fixture(path):
    """the reference's own per-rank outputs (it ran with W ranks over gloo on its CPU path) through the IPC transport"""
    import numpy as np
    world = int(np.load(path)["meta"][0])
    _run_ranks(_fixture_worker, world, (path, "ipc"), timeout=300)


@pytest.mark.parametrize("tokens", [[512, 0], [0, 384], [512, 200]], ids=lambda t: "x".join(map(str, t)))
def test_ipc_transport_unequal_or_empty_ranks(tokens):
    """a rank without tokens still stores its (all-zero) bucket rows into every peer and signals every stage"""
    cfg = dict(shape=(512, 128, 192, 2), E_loc=2, dtype="bfloat16", sweep=[(1, 1), (1, 2)], tokens=tokens, transport="ipc")
    _run_ranks(_sweep_worker, 2, (cfg,), timeout=300)


def test_ipc_transport_gates_applied_in_encode():
    """is_postscore=False (fast_dispatch.py:125: the gate multiplies the token on its way INTO the bucket): the peer-store encode takes its
    gated branch, decode adds plain rows; two ranks, degrees 1 and 2, bf16"""
    cfg = dict(shape=(512, 128, 192, 2), E_loc=4, dtype="bfloat16", sweep=[(1, 1), (1, 2)], transport="ipc", postscore=False)
    _run_ranks(_sweep_worker, 2, (cfg,), timeout=300)


def test_ipc_transport_degree_sweep_keeps_the_bits():
    """degrees 1..8 over one layer (fp16, 16 local experts): every output vs the oracle at 1e-3 and bit-identical to degree 1
    whenever the stages keep the rows per launch"""
    cfg = dict(shape=(13440, 256, 512, 2), E_loc=16, dtype="float16", sweep=[(1, o) for o in (1, 2, 3, 4, 7, 8)], transport="ipc")
    _run_ranks(_sweep_worker, 2, (cfg,))


def _graph_worker(rank, world, port, replays, degree, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        import time
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd import _lib
        from tutel_amd.impls import ep_native
        from tutel_amd.impls.graph import GraphedForward
        _set_transport(ep_native, "ipc")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        T, M, H, k, E_loc = 512, 256, 256, 2, 4
        E = E_loc * world
        dtype = torch.bfloat16
        x = O.make_problem(T, M, H, E, dtype=dtype, seed=100 + rank)[0].cuda()
        x2 = O.make_problem(T, M, H, E, dtype=dtype, seed=300 + rank)[0].cuda()
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        layer = moe.moe_layer(gate_type={"type": "top", "k": k, "fp32_gate": True}, model_dim=M,
                              experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)},
                              seeds=(1, rank + 1, 1), a2a_ffn_overlap_degree=degree)
        torch.set_default_dtype(old)
        layer = layer.cuda().eval()
        with torch.no_grad():
            want, want2 = layer(x).clone(), layer(x2).clone()
        torch.cuda.synchronize()
        graphed = GraphedForward(layer, x)
        comm = ep_native.communicator(layer.group, torch.device("cuda", 0))
        assert comm.ipc
        ok, t0 = True, time.time()
        for i in range(replays):
            src, ref = (x2, want2) if i % 7 == 3 else (x, want)   # the static input changes now and then: the replay is a real forward
            y = graphed(src)
            if i % 13 == 0 or i == replays - 1:
                torch.cuda.synchronize()
                ok = ok and torch.equal(y, ref)
        torch.cuda.synchronize()
        dt = time.time() - t0
        _lib.check(_lib.lib().tutel_amd_ep_ipc_status(comm.handle), "tutel_amd_ep_ipc_status")
        # eager after the graph: the epoch counters on the device moved with the replays, the eager path follows them
        with torch.no_grad():
            ok = ok and torch.equal(layer(x), want)
        torch.cuda.synchronize()
        q.put((rank, bool(ok), f"{replays} replays in {dt:.2f} s ({dt / replays * 1e6:.0f} us each)", []))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


def _stress_worker(rank, world, port, degree, n_forwards, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd import _lib
        from tutel_amd.impls import ep_native
        _set_transport(ep_native, "ipc")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        T, M, H, k, E_loc = 384, 128, 128, 2, 4
        E = E_loc * world
        dtype = torch.bfloat16
        batches = [O.make_problem(T, M, H, E, dtype=dtype, seed=500 + 10 * b + rank)[0].cuda() for b in range(3)]
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        layer = moe.moe_layer(gate_type={"type": "top", "k": k, "fp32_gate": True}, model_dim=M,
                              experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)},
                              seeds=(1, rank + 1, 1), a2a_ffn_overlap_degree=degree)
        torch.set_default_dtype(old)
        layer = layer.cuda().eval()
        with torch.no_grad():
            want = [layer(b).clone() for b in batches]
            torch.cuda.synchronize()
            outs = []
            for i in range(n_forwards):     # no host synchronisation in between: the ranks drift apart as far as the protocol lets them
                b = (i * 7 + i // 5) % 3
                outs.append((b, layer(batches[b])))
                if len(outs) == 50:
                    torch.cuda.synchronize()
                    bad = [j for j, (bb, y) in enumerate(outs) if not torch.equal(y, want[bb])]
                    assert not bad, f"forward {i - 49 + bad[0]}: the result of batch {outs[bad[0]][0]} changed"
                    outs = []
        torch.cuda.synchronize()
        comm = ep_native.communicator(layer.group, torch.device("cuda", 0))
        _lib.check(_lib.lib().tutel_amd_ep_ipc_status(comm.handle), "tutel_amd_ep_ipc_status")
        q.put((rank, True, f"{n_forwards} eager forwards over 3 batches", []))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


@pytest.mark.parametrize("n_forwards", [100, pytest.param(600, marks=pytest.mark.slow)])
@pytest.mark.parametrize("world,degree", [(2, 2), (4, 2), (4, 1)])
def test_ipc_transport_eager_stress_with_changing_batches(world, degree, n_forwards):
    """eager forwards without host synchronisation (100 in the default run, 600 with --runslow), the batch changing from call to call:
    every result must equal, bit for bit, the first result of its batch -- a bucket row left over from another batch (a flag seen too
    early, a buffer reused too soon) would show"""
    _run_ranks(_stress_worker, world, (degree, n_forwards), timeout=600)


@pytest.mark.parametrize("replays", [200, pytest.param(2000, marks=pytest.mark.slow)])
@pytest.mark.parametrize("degree", [1, 2])
def test_ipc_transport_graph_replays(degree, replays):
    """replaying captured RCCL collectives hung after ~200 replays.  The IPC transport is plain kernels + events with
    its epochs counted in device memory: replays of the captured two-rank forward (200 by default, 2000 with --runslow), the static
    input rewritten now and then, outputs checked along the way"""
    _run_ranks(_graph_worker, 2, (replays, degree), timeout=600)


def _timeout_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd import _lib
        from tutel_amd.impls import ep_native
        _set_transport(ep_native, "ipc")
        ep_native.IPC_TIMEOUT_MS = 400
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        T, M, H, k, E_loc = 256, 128, 128, 2, 2
        dtype = torch.bfloat16
        x = O.make_problem(T, M, H, E_loc * world, dtype=dtype, seed=100 + rank)[0].cuda()
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        layer = moe.moe_layer(gate_type={"type": "top", "k": k, "fp32_gate": True}, model_dim=M,
                              experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)}, seeds=(1, rank + 1, 1))
        torch.set_default_dtype(old)
        layer = layer.cuda().eval()
        with torch.no_grad():
            layer(x)                      # both ranks: communicator, segments, one good forward
        torch.cuda.synchronize()
        dist.barrier()
        ok, info = True, ""
        if rank == 0:  # noqa: SIM102
            with torch.no_grad():
                y_dead = layer(x)         # rank 1 never makes this call: the wait kernel gives up after 0.4 s ...
            torch.cuda.synchronize()      # ... so this returns
            # ... and what it hands back is not a result: poisoned with NaN on the device (it used to be garbage + status 0)
            ok = bool(torch.isnan(y_dead.float()).all())
            info = "" if ok else "the output of a timed-out forward must be NaN"
            try:
                ep_native.ipc_status(layer.group)      # and the error can be asked for without making another call
                ok, info = False, "ipc_status must raise after a timed-out exchange"
            except _lib.TutelAmdError:
                pass
            try:
                with torch.no_grad():
                    layer(x)              # ... and the next call reports who never arrived
                ok, info = False, "the forward after a timed-out exchange must raise"
            except _lib.TutelAmdError as ex:
                info = info or str(ex)
                ok = ok and "timed out waiting for rank 1" in str(ex)
        dist.barrier()
        q.put((rank, bool(ok), info, []))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


def test_ipc_transport_bounded_wait_reports_the_missing_peer():
    _run_ranks(_timeout_worker, 2, (), timeout=300)


def _capacity_worker(rank, world, port, q):
    """with the IPC transport the capacity bucket of a workspace decides WHICH peer-mapped segment the kernels
    store into.  Ranks with different token counts keep different workspace caches, so the bucket must come from the call's
    (agreed) capacity alone: here the capacity changes from call to call while the ranks' token counts differ -- a rank that
    re-used a larger cached workspace would store into another segment (or `back` offset) than its peer reads."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd.impls import ep_native
        _set_transport(ep_native, "ipc")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        Tmax, M, H, k, E_loc = 1024, 128, 192, 2, 2
        E = E_loc * world
        dtype = torch.bfloat16
        full = [O.make_problem(Tmax, M, H, E, dtype=dtype, seed=100 + r)[0] for r in range(world)]
        _, wg, w1, b1, w2, b2 = O.make_problem(Tmax, M, H, E, dtype=dtype, seed=7)
        sl = slice(rank * E_loc, (rank + 1) * E_loc)
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        layer = moe.moe_layer(gate_type={"type": "top", "k": k, "fp32_gate": True}, model_dim=M,
                              experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)})
        torch.set_default_dtype(old)
        with torch.no_grad():
            layer.gates[0].wg.weight.copy_(wg.float())
            layer.experts.batched_fc1_w.copy_(w1[sl]); layer.experts.batched_fc1_bias.copy_(b1[sl])
            layer.experts.batched_fc2_w.copy_(w2[sl]); layer.experts.batched_fc2_bias.copy_(b2[sl])
        layer = layer.cuda().eval()
        parts = lambda t: [t[r * E_loc:(r + 1) * E_loc] for r in range(world)]
        # (tokens of rank 0, tokens of rank 1): the agreed capacity goes 500 -> 250 -> 500 -> 126 -> 500 while the ranks swap sizes
        calls = [(100, 1000), (200, 500), (100, 1000), (250, 60), (1000, 100), (200, 500)]
        ok, report, buckets = True, [], []
        for degree in (1, 2):
            for toks in calls:
                xs = [full[r][:toks[r]] for r in range(world)]
                with torch.no_grad():
                    y = layer(xs[rank].cuda(), a2a_ffn_overlap_degree=degree, inequivalent_tokens=True)
                torch.cuda.synchronize()
                box = [None]
                if rank == 0:
                    box[0] = O.moe_forward_ep(xs, wg, parts(w1), parts(b1), parts(w2), parts(b2), top_k=k, fp32_gate=True,
                                              alignment=degree, accum_fp32=True, inequivalent_tokens=True)
                dist.broadcast_object_list(box, src=0)
                want, crits = box[0]
                err = (y.cpu().double() - want[rank].double()).abs()
                scale = float(want[rank].double().abs().max())
                tol = 2 ** -7 * want[rank].double().abs() + max(2e-3, 2 ** -8 * scale)
                bad = int((err > tol).sum())
                ok = ok and bad == 0 and torch.equal(layer.dispatch_count.cpu(), crits[rank][5])
                report.append(f"o={degree} tokens {toks} capacity {crits[rank][4]}: max err {float(err.max()):.2e}, {bad} over the bar")
                buckets.append(sorted({w.C_cap for w in layer._ep_workspaces.values()}))
        comm = ep_native.communicator(layer.group, torch.device("cuda", 0))
        assert comm is not None and comm.ipc
        # every rank went through the same sequence of segments (their keys carry the capacity bucket)
        keys = [None] * world
        dist.all_gather_object(keys, sorted(str(kk) for kk in comm.segments))
        ok = ok and all(kk == keys[0] for kk in keys) and len(keys[0]) >= 3
        q.put((rank, bool(ok), "; ".join(report) + f"; segments {keys[0]}", []))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


def test_ipc_transport_capacity_changes_between_calls_with_unequal_ranks():
    _run_ranks(_capacity_worker, 2, (), timeout=300)


def _stale_worker(rank, world, port, q):
    """the epoch canaries: rank 1 publishes the PREVIOUS epoch behind its rows (TUTEL_OPT_EP_CANARY = 2: as if its stores had not
    landed when its flag did).  Every rank that consumes rank 1's rows must notice in its wait kernel, poison its output and fail
    its next call with a text that names rank 1 and says what happened."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd import _lib, ops
        from tutel_amd.impls import ep_native
        _set_transport(ep_native, "ipc")
        ep_native.IPC_TIMEOUT_MS = 20000
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        T, M, H, k, E_loc = 256, 128, 128, 2, 2
        dtype = torch.bfloat16
        x = O.make_problem(T, M, H, E_loc * world, dtype=dtype, seed=100 + rank)[0].cuda()
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        layer = moe.moe_layer(gate_type={"type": "top", "k": k, "fp32_gate": True}, model_dim=M,
                              experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                       "activation_fn": lambda t: torch.nn.functional.relu(t)}, seeds=(1, rank + 1, 1))
        torch.set_default_dtype(old)
        layer = layer.cuda().eval()
        with torch.no_grad():
            good = [layer(x, a2a_ffn_overlap_degree=d).clone() for d in (1, 2, 1)]   # canaries on: three good forwards
        torch.cuda.synchronize()
        ep_native.ipc_status(layer.group)
        ok = bool(torch.isfinite(good[0].float()).all()) and torch.equal(good[0], good[2])
        dist.barrier()
        if rank == 1:
            ops.set_option(_lib.OPT_EP_CANARY, 2)
        with torch.no_grad():
            y = layer(x)
        torch.cuda.synchronize()
        info = ""
        ok = ok and bool(torch.isnan(y.float()).all())     # both ranks consumed a block of rank 1 (its own included)
        try:
            with torch.no_grad():
                layer(x)
            ok, info = False, "the forward after a stale exchange must raise"
        except _lib.TutelAmdError as ex:
            info = str(ex)
            ok = ok and "rank 1's flag BEFORE the rows" in info and "TUTEL_AMD_EP_TRANSPORT=rccl" in info
        dist.barrier()
        q.put((rank, bool(ok), info, []))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


def test_ipc_transport_epoch_canaries_report_rows_behind_their_flag():
    _run_ranks(_stale_worker, 2, (), timeout=300)


def _selfcheck_worker(rank, world, port, q):
    """the payload-sized self-check itself: every store flavour x both stream layouts between the rank processes, mismatches
    counted on the device; and what the attach-time check recorded"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from tutel_amd.impls import ep_native
        _set_transport(ep_native, "ipc")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        comm = ep_native.communicator(None, dev)
        assert comm is not None and comm.ipc
        sc = comm.selfcheck
        ok = sc["mismatches"] == [0, 0] and sc["bytes_per_peer"] * world >= 32 << 20 and sc["passes"] >= 4
        seg = ep_native._open_segment(comm, world * (8 << 20), False)
        report = []
        for flavour in (0, 1, 2, 3):
            for side in (False, True):
                bad, _ = ep_native.ipc_selfcheck(comm, seg, 8 << 20, 6, flavour, side)
                report.append(bad)
                if flavour == 0:            # plain stores are what the pipeline uses: must be exact.  The other flavours are probes
                    ok = ok and bad == 0    # (round 4 saw stale rows with write-through stores): reported, not asserted
        q.put((rank, bool(ok), f"attach check {sc}; mismatches per (flavour, layout): {report}", []))
        dist.barrier()
        ep_native.destroy_all()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


def _segment_retry_worker(rank, world, port, q):
    """a rank whose first allocation / export fails (seen once as a transient "invalid argument" with three processes on one device):
    every rank undoes its attempt and repeats it, the transport attaches on the second try and carries a payload-sized check; a rank
    that fails EVERY attempt makes every rank give the transport up, with one error, not a hang"""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from tutel_amd import _lib
        from tutel_amd.impls import ep_native
        _set_transport(ep_native, "ipc")
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        dev = torch.device("cuda", 0)
        ep_native._SEGMENT_FAULTS = {(world - 1, 0)}            # the last rank's first attempt of EVERY segment
        comm = ep_native.communicator(None, dev)
        ok = comm is not None and comm.ipc and comm.selfcheck["mismatches"] == [0, 0]
        seg = ep_native._open_segment(comm, world * (4 << 20), False)
        bad, _ = ep_native.ipc_selfcheck(comm, seg, 4 << 20, 3, 0, False)
        ok = ok and bad == 0
        ep_native._SEGMENT_FAULTS = {(0, a) for a in range(ep_native.SEGMENT_ATTEMPTS)}   # rank 0 never succeeds
        gave_up = False
        try:
            ep_native._open_segment(comm, 1 << 20, False)
        except _lib.TutelAmdError:
            gave_up = True
        ep_native._SEGMENT_FAULTS = set()
        q.put((rank, bool(ok and gave_up), f"attached after a retry: {ok}; every rank gave up together: {gave_up}", []))
        dist.barrier()
        ep_native.destroy_all()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_segment_attach_retries_a_failed_attempt_collectively(world):
    _run_ranks(_segment_retry_worker, world, (), timeout=300)


@pytest.mark.parametrize("world", [2, 3])
def test_ipc_transport_payload_sized_selfcheck(world):
    res = _run_ranks(_selfcheck_worker, world, (), timeout=300)
    print(res[0][2])


@pytest.mark.parametrize("shape,n_eager,n_replays", [((1024, 512, 256, 2, 4), 120, 100), ((4096, 2048, 1024, 2, 8), 60, 60)],
                         ids=["small", "16.8MB-per-peer"])
def test_multi_gpu_stress_worker_with_ranks_sharing_one_gpu(shape, n_eager, n_replays):
    """the worker tests/test_multi_gpu_rccl.py runs between real GPUs (changing batches, no host synchronisation, replays with the
    static input rewritten, every output against the oracle), here with the two ranks on the one GPU: the same code, so that the
    first multi-GPU run does not also debug its own test"""
    from test_multi_gpu_rccl import _stress_worker as multi_gpu_stress
    _run_ranks(multi_gpu_stress, 2, ("ipc", shape, n_eager, n_replays, True), timeout=600)
