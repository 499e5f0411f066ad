"""Expert parallelism between REAL GPUs (backend "nccl" rendezvous, one process per GPU), over BOTH transports of the library:
"ipc" -- the round-4 exchange, peer stores over xGMI from fast_encode and the fc2 epilogue, flag kernels, no collective -- and
"rccl" -- ncclAllToAll on the library communicator (the fallback when a rank cannot map its peers' segments).

Auto-enabled when the box exposes >= 2 GPUs (skipped on single-GPU boxes): W = 2 (and 4 / 8 when the
devices are there), a2a_ffn_overlap_degree 1 and 2, against the oracle's W-rank simulation -- the
reference's own assertion for this path is overlap-invariance (tests/test_tutel.py:161-176), checked
here as well (degree 2 output == degree 1 output, bitwise) -- plus, on the IPC transport, 300 HIP-graph replays.
Round 5: test_changing_batch_stress_between_gpus -- three different batches at >= 16 MB per peer, hundreds of eager
forwards without host synchronisation and graph replays with the static input rewritten, every output checked; the very same
worker runs on a 1-GPU box with the ranks sharing the device (tests/test_ep_ipc_one_gpu.py)."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

pytestmark = pytest.mark.gpu


def _ngpu():
    try:
        return torch.cuda.device_count() if torch.cuda.is_available() else 0
    except Exception:
        return 0


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, E_loc, transport, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK=str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", TUTEL_AMD_EP_TRANSPORT=transport)
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd.impls import ep_native
        assert ep_native.TRANSPORT == transport
        torch.cuda.set_device(rank)
        dev = torch.device("cuda", rank)
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        T, M, H, k = 1024, 256, 512, 2
        E = E_loc * world
        dtype = torch.bfloat16
        xs = [O.make_problem(T, M, H, E, dtype=dtype, seed=100 + r)[0] for r in range(world)]
        _, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=dtype, seed=7)
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
        layer = layer.to(dev).eval()
        assert layer.world_size == world and layer.num_global_experts == E
        x = xs[rank].to(dev)
        outs = {}
        with torch.no_grad():
            for degree in (1, 2, 1, 2):   # repeated: buffer reuse across the two streams must stay safe
                outs.setdefault(degree, []).append(layer(x, a2a_ffn_overlap_degree=degree).clone())
        torch.cuda.synchronize()
        comm = ep_native.communicator(layer.group, dev)
        assert comm is not None and comm.ipc == (transport == "ipc"), f"transport {transport}: communicator ipc={getattr(comm, 'ipc', None)}"
        if transport == "ipc":   # the captured forward replays (kernels + events only; RCCL collectives are refused by GraphedForward)
            from tutel_amd.impls.graph import GraphedForward
            with torch.no_grad():
                gf = GraphedForward(layer, x, a2a_ffn_overlap_degree=2)
                for _ in range(300):
                    yg = gf(x)
            torch.cuda.synchronize()
            outs[2].append(yg.clone())
        # capacity = k * ceil(T/E) is even here, so the alignment rule (moe_layer.py:298-301) leaves it alone
        want, crits = O.moe_forward_ep(xs, wg, [w1[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       [b1[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       [w2[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       [b2[r * E_loc:(r + 1) * E_loc] for r in range(world)],
                                       top_k=k, fp32_gate=True, alignment=2, accum_fp32=True)
        y = outs[1][0]
        err = (y.cpu().double() - want[rank].double()).abs()
        tol = 2 ** -7 * want[rank].double().abs() + 2e-3
        ok = bool((err <= tol).all()) and torch.equal(layer.dispatch_count.cpu(), crits[rank][5])
        same = all(torch.equal(y, o) for d in outs for o in outs[d])
        # variable-size collectives (net.batch_all_to_all_v / batch_all_gather_v, custom_kernel.cpp:463-518) on the library's
        # communicator: ragged seeded splits incl. an empty pair
        from tutel import net
        gen = torch.Generator().manual_seed(5)
        split = torch.randint(0, 4000, [world, world], generator=gen)
        split[0, world - 1] = 0
        datas = [(torch.randn(int(split[s].sum()), generator=gen) * 100).to(dtype) for s in range(world)]
        (got,), rs = net.batch_all_to_all_v([datas[rank].to(dev)], split[rank].tolist())
        offs = [[int(split[s, :d].sum()) for d in range(world)] for s in range(world)]
        exp = torch.cat([datas[s][offs[s][rank]:offs[s][rank] + int(split[s, rank])] for s in range(world)])
        vok = torch.equal(got.cpu(), exp) and rs.cpu().tolist() == split[:, rank].tolist()
        (gg,), _ = net.batch_all_gather_v([datas[rank].to(dev)])
        vok = vok and torch.equal(gg.cpu(), torch.cat(datas))
        # a rank WITHOUT tokens must not hang its peers (inequivalent_tokens)
        with torch.no_grad():
            xe = x if rank != world - 1 else x[:0]
            ye = layer(xe, a2a_ffn_overlap_degree=2, inequivalent_tokens=True)
        torch.cuda.synchronize()
        vok = vok and ye.shape[0] == xe.shape[0] and bool(torch.isfinite(ye.float()).all())
        q.put((rank, ok and same and vok, f"max err {float(err.max()):.3e}; degree 1 == degree 2 bitwise: {same}; v-collectives + empty rank: {vok}"))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc()))


def _stress_worker(rank, world, port, transport, shape, n_eager, n_replays, share, q):
    """what the same-`x` test above cannot see.  Three DIFFERENT batches per rank at a payload of >= 16 MB per
    peer and direction (shape), each first forward compared with the oracle; then n_eager forwards WITHOUT host synchronisation with the
    batch changing from call to call, then n_replays replays of the captured forward with the static input REWRITTEN between the
    replays (IPC transport) -- every single output compared, bit for bit, with the first result of its batch, i.e. transitively with
    the oracle.  A row of the batch in between (a flag that overtook its rows, a buffer reused too soon, a stale line) cannot hide
    behind identical inputs.  share=True: the ranks share cuda:0 over a gloo rendezvous (how a 1-GPU box runs this very worker,
    tests/test_ep_ipc_one_gpu.py), otherwise one GPU per rank over nccl."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world),
                          LOCAL_RANK="0" if share else str(rank), HSA_ENABLE_IPC_MODE_LEGACY="0", TUTEL_AMD_EP_TRANSPORT=transport)
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd.impls import ep_native
        from tutel_amd.impls.graph import GraphedForward
        ep_native.TRANSPORT = transport
        dev = torch.device("cuda", 0 if share else rank)
        torch.cuda.set_device(dev)
        if share:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        T, M, H, k, E_loc = shape
        E = E_loc * world
        dtype = torch.bfloat16
        nb = 3
        xs = [[O.make_problem(T, M, H, E, dtype=dtype, seed=1000 + 10 * b + r)[0] for r in range(world)] for b in range(nb)]
        _, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=dtype, seed=7)
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
        layer = layer.to(dev).eval()
        mine = [xs[b][rank].to(dev) for b in range(nb)]
        parts = lambda t: [t[r * E_loc:(r + 1) * E_loc] for r in range(world)]
        report, ok = [], True
        per_peer_mb = E_loc * (k * ((T + E - 1) // E)) * M * 2 / 1e6
        for degree in (1, 2):
            with torch.no_grad():
                want = [layer(mine[b], a2a_ffn_overlap_degree=degree).clone() for b in range(nb)]
            torch.cuda.synchronize()
            comm = ep_native.communicator(layer.group, dev)
            assert comm is not None and comm.ipc == (transport == "ipc"), f"transport {transport}: communicator ipc={getattr(comm, 'ipc', None)}"
            for b in range(nb):      # every batch's first result against the oracle (rank 0 computes, everybody checks its own part)
                box = [None]
                if rank == 0:
                    box[0] = O.moe_forward_ep(xs[b], wg, parts(w1), parts(b1), parts(w2), parts(b2), top_k=k, fp32_gate=True,
                                              alignment=degree, accum_fp32=True)[0]
                dist.broadcast_object_list(box, src=0, device=dev if not share else None)
                ref = box[0][rank]
                err = (want[b].cpu().double() - ref.double()).abs()
                scale = float(ref.double().abs().max())
                bad = int((err > 2 ** -7 * ref.double().abs() + max(2e-3, 2 ** -8 * scale)).sum())
                ok = ok and bad == 0
                report.append(f"o={degree} batch {b}: max err {float(err.max()):.2e} ({bad} over the bar)")
            with torch.no_grad():
                outs = []
                for i in range(n_eager):     # no host synchronisation in between: the ranks drift apart as far as the protocol lets them
                    b = (i * 7 + i // 5) % nb
                    outs.append((b, layer(mine[b], a2a_ffn_overlap_degree=degree)))
                    if len(outs) == 50 or i == n_eager - 1:
                        torch.cuda.synchronize()
                        wrong = [j for j, (bb, y) in enumerate(outs) if not torch.equal(y, want[bb])]
                        assert not wrong, f"degree {degree}, eager forward {i - len(outs) + 1 + wrong[0]}: the result of batch {outs[wrong[0]][0]} changed"
                        outs = []
            ep_native.ipc_status()
            if transport == "ipc" and n_replays:   # the captured forward, its static input rewritten between replays
                with torch.no_grad():
                    gf = GraphedForward(layer, mine[0], a2a_ffn_overlap_degree=degree)
                    outs = []
                    for i in range(n_replays):
                        b = (i * 5 + i // 3) % nb
                        outs.append((b, gf(mine[b]).clone()))
                        if len(outs) == 50 or i == n_replays - 1:
                            torch.cuda.synchronize()
                            wrong = [j for j, (bb, y) in enumerate(outs) if not torch.equal(y, want[bb])]
                            assert not wrong, f"degree {degree}, replay {i - len(outs) + 1 + wrong[0]}: the result of batch {outs[wrong[0]][0]} changed"
                            outs = []
                del gf
                ep_native.ipc_status()
            report.append(f"o={degree}: {n_eager} eager forwards + {n_replays if transport == 'ipc' else 0} replays over {nb} batches, bitwise")
        q.put((rank, bool(ok), f"{per_peer_mb:.1f} MB per peer and direction; " + "; ".join(report), []))
        dist.barrier()
        ep_native.destroy_all()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


@pytest.mark.parametrize("transport", ["ipc", "rccl"])
def test_changing_batch_stress_between_gpus(transport):
    """two GPUs, 16.8 MB per peer and direction (T = 4096, M = 2048, 8 local experts, capacity 512): 3 batches vs the oracle,
    240 eager forwards without host synchronisation + 300 graph replays with the static input rewritten, degrees 1 and 2"""
    world = 2
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_ngpu()}")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_stress_worker, args=(r, world, port, transport, (4096, 2048, 1024, 2, 8), 240, 300, False, q)) for r in range(world)]
    from test_ep_ranks_one_gpu import _rank_env
    with _rank_env(world, share_gpu=False):   # the host's cores shared out between the ranks' OpenMP pools
        for p in procs:
            p.start()
    res = [q.get(timeout=900) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info, _ in res:
        assert ok, f"rank {rank}: {info}"


@pytest.mark.parametrize("transport", ["ipc", "rccl"])
@pytest.mark.parametrize("world,E_loc", [(2, 4), (2, 3), (4, 2), (8, 8)])
def test_expert_parallel_between_gpus(world, E_loc, transport):
    if _ngpu() < world:
        pytest.skip(f"needs {world} GPUs, this box has {_ngpu()}")
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, E_loc, transport, q)) for r in range(world)]
    from test_ep_ranks_one_gpu import _rank_env
    with _rank_env(world, share_gpu=False):   # the host's cores shared out between the ranks' OpenMP pools
        for p in procs:
            p.start()
    res = [q.get(timeout=600) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, ok, info in res:
        assert ok, f"rank {rank}: {info}"
