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
import torch.multiprocessing as mp

from test_ep_ranks_one_gpu import (_ep_fixtures, _fixture_worker, _free_port, _rank_env, _run_ranks, _set_transport, _sweep_worker, _worker)

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("world,degree,E_loc", [(2, 1, 2), (2, 2, 4), (2, 2, 3), (2, 4, 4), (4, 2, 2), (4, 1, 1)])
def test_ipc_transport_ranks_sharing_one_gpu(world, degree, E_loc):
    """expert-sliced and capacity-chunked stages, degrees 1 / 2 / 4: vs the oracle, three calls bit-identical, and bit-identical
    to the host-staged exchange of the same forward"""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, degree, E_loc, q, None, "ipc")) for r in range(world)]
    with _rank_env(world):
        for p in procs:
            p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info, plans in res:
        assert ok, f"rank {rank}: {info}"
        if degree > 1:
            assert plans == [E_loc % degree == 0]


def test_ipc_transport_rank_shape_with_256_row_tiles():
    """8 local experts x 1024 rows, M = H = 2048 (the per-rank problem of an 8-way run at the headline dims, reproduced with two
    ranks): the stage GEMMs run the 256-row-tile kernels, whose epilogue goes through LDS -- that epilogue's peer stores"""
    world, degree, E_loc = 2, 2, 8
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, degree, E_loc, q, (4096, 2048, 2048, 2), "ipc")) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info, plans in res:
        assert ok, f"rank {rank}: {info}"


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


@pytest.mark.parametrize("world,degree", [(2, 2), (4, 2), (4, 1)])
def test_ipc_transport_eager_stress_with_changing_batches(world, degree):
    """600 eager forwards without host synchronisation, the batch changing from call to call: every result must equal, bit for bit,
    the first result of its batch -- a bucket row left over from another batch (a flag seen too early, a buffer reused too soon)
    would show"""
    _run_ranks(_stress_worker, world, (degree, 600), timeout=600)


@pytest.mark.parametrize("degree", [1, 2])
def test_ipc_transport_2000_graph_replays(degree):
    """replaying captured RCCL collectives hung after ~200 replays.  The IPC transport is plain kernels + events with
    its epochs counted in device memory: 2000 replays of the captured two-rank forward, outputs checked along the way"""
    _run_ranks(_graph_worker, 2, (2000, degree), timeout=600)


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
        if rank == 0:
            with torch.no_grad():
                layer(x)                  # rank 1 never makes this call: the wait kernel gives up after 0.4 s ...
            torch.cuda.synchronize()      # ... so this returns
            try:
                with torch.no_grad():
                    layer(x)              # ... and the next call reports who never arrived
                ok, info = False, "the forward after a timed-out exchange must raise"
            except _lib.TutelAmdError as ex:
                info = str(ex)
                ok = "timed out waiting for rank 1" in info
        dist.barrier()
        q.put((rank, bool(ok), info, []))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


def test_ipc_transport_bounded_wait_reports_the_missing_peer():
    _run_ranks(_timeout_worker, 2, (), timeout=300)
