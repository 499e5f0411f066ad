"""Generates tests/golden/ep_*.npz by running the REFERENCE itself with W > 1 ranks (microsoft/tutel from /root/reference on its
own C++ CPU kernels in oracle/_ref/, one process per rank over gloo): the expert-parallel forward -- routing per rank, the
all-to-all, experts on the received [E_loc, W*C, M] rows, the all-to-all back, decode -- incl. `inequivalent_tokens=True` with ranks
holding different numbers of tokens.  Only runs in the build container; the fixtures are committed.

    make -C oracle && python tests/golden/make_golden_ep.py [--check]

Inputs are regenerated from seeds by oracle.moe_oracle.make_problem (rank r's tokens: seed 100 + r, weights: seed 7, rank r owns
experts [r*E_loc, (r+1)*E_loc)), so a fixture holds the reference's OUTPUTS per rank: y, the rows its experts received after the
all-to-all, dispatch_count, capacity, l_aux.  --check compares oracle.moe_forward_ep with the live reference instead of writing."""
import argparse
import os
import socket
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
REF = os.environ.get("TUTEL_REFERENCE", "/root/reference")

# (name, world, T, M, H, E_loc, k, dtype, fp32_gate, tokens per rank | None, capacity_factor)
CASES = [
    ("w2_f32_k2", 2, 256, 64, 32, 2, 2, "float32", False, None, 1.0),
    ("w4_f32_k1_e1", 4, 192, 64, 32, 1, 1, "float32", False, None, 1.0),
    ("w2_f32_k2_unequal", 2, 256, 64, 32, 2, 2, "float32", False, [256, 100], 1.0),
    ("w2_f32_k2_unequal_drop", 2, 256, 64, 32, 2, 2, "float32", False, [90, 256], 0.5),
    ("w2_bf16_k2_fp32gate", 2, 256, 64, 64, 4, 2, "bfloat16", True, None, 1.0),
    ("w2_f32_k2_dropless", 2, 256, 64, 32, 2, 2, "float32", False, None, 0.0),
]


# sharded experts (one expert's hidden dim sliced over the ranks, num_experts_per_device = -W): the three
# parallel modes the reference compares with each other (tests/test_tutel.py:154-159) -- weights from the layer's own seeds
SHARDED = ("sharded_w2_f32_k1", 2, 128, 32, 16, 1)   # name, W, T, M, H, k


def _sharded_worker(rank, world, port, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        sys.path = [REF, os.path.join(ROOT, "oracle", "_ref")] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT] + [ROOT]
        import logging
        import torch
        import torch.distributed as dist
        logging.disable(logging.CRITICAL)
        import tutel
        assert os.path.abspath(tutel.__file__).startswith(REF), tutel.__file__
        from tutel import moe as ref_moe
        dist.init_process_group("gloo", rank=rank, world_size=world)
        name, W, T, M, H, k = SHARDED
        outs = {}
        for ptype in ("data", "model", "adaptive:0"):
            layer = ref_moe.moe_layer(gate_type={"type": "top", "k": k},
                                      experts={"type": "ffn", "num_experts_per_device": -W, "hidden_size_per_expert": H,
                                               "activation_fn": lambda t: torch.nn.functional.relu(t)},
                                      model_dim=M, parallel_type=ptype, seeds=(1, rank + 1, 1)).eval()
            torch.manual_seed(0)
            x = torch.randn(T, M)
            with torch.no_grad():
                outs[ptype] = layer(x)
        q.put((rank, {k: v.numpy().copy() for k, v in outs.items()}, True, None, None))   # by value: the worker may exit before the parent reads
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None, None, None))


def run_sharded_reference():
    import torch.multiprocessing as mp
    W = SHARDED[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_sharded_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[2] is not None, r[1]
    return res


def _net_worker(rank, world, port, q):
    """tutel.net of the REFERENCE and of this repo (tutel_amd.net) side by side in one process per rank, gloo: the collectives with
    and without autograd, same inputs -> equal outputs and equal gradients.  Calls the reference cannot run on a gloo group are
    listed, not compared."""
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        sys.path = [REF, os.path.join(ROOT, "oracle", "_ref")] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT] + [ROOT]
        import logging
        import torch
        import torch.distributed as dist
        logging.disable(logging.CRITICAL)
        import tutel
        assert os.path.abspath(tutel.__file__).startswith(REF), tutel.__file__
        from tutel import net as rnet
        from tutel_amd import net as mnet
        dist.init_process_group("gloo", rank=rank, world_size=world)
        g = torch.Generator().manual_seed(40 + rank)
        bad, skipped = [], []

        def both(name, fn, *tensors, grad=False):
            outs = []
            for net in (rnet, mnet):
                ins = [t.clone().requires_grad_(grad) for t in tensors]
                try:
                    y = fn(net, *ins)
                except Exception as ex:   # noqa: BLE001
                    if net is rnet:
                        skipped.append(f"{name}: reference raises on gloo ({type(ex).__name__})")
                        return
                    bad.append(f"{name}: product raises {type(ex).__name__}: {ex}")
                    return
                ys = list(y) if isinstance(y, (tuple, list)) else [y]
                ys = [t for t in ys if torch.is_tensor(t)]
                gr = []
                if grad:
                    sum((t.float() * torch.arange(t.numel(), dtype=torch.float32).view(t.shape)).sum() for t in ys).backward()
                    gr = [i.grad for i in ins]
                outs.append((ys, gr))
            (yr, gr_), (ym, gm) = outs
            if len(yr) != len(ym) or not all(a.shape == b.shape and torch.equal(a, b) for a, b in zip(yr, ym)):
                bad.append(f"{name}: outputs differ")
            if grad and not all((a is None and b is None) or (a is not None and b is not None and torch.allclose(a, b)) for a, b in zip(gr_, gm)):
                bad.append(f"{name}: gradients differ")
        x = torch.randn([4 * world, 6, 8], generator=g)
        both("simple_all_reduce", lambda n, t: n.simple_all_reduce(t), x)
        both("simple_all_to_all", lambda n, t: n.simple_all_to_all(t), x)
        both("simple_split", lambda n, t: n.simple_split(t), x)
        both("simple_all_gather", lambda n, t: n.simple_all_gather(t), x)
        both("simple_reduce_scatter", lambda n, t: n.simple_reduce_scatter(t), x)
        for i, o in ((0, 1), (1, 0), (0, 2), (2, 0), (1, 2), (0, 0)):
            xx = torch.randn([2 * world, 3 * world, 4 * world], generator=g)
            both(f"all_to_all({i},{o})", lambda n, t, i=i, o=o: n.all_to_all(t, i, o), xx, grad=True)
            both(f"all_to_all({i},{o},use_2dh)", lambda n, t, i=i, o=o: n.all_to_all(t, i, o, use_2dh=True), xx, grad=True)
        both("all_to_all_single", lambda n, t: n.all_to_all_single(t), x, grad=True)
        for d in (0, 1, 2):
            both(f"all_gather(dim={d})", lambda n, t, d=d: n.all_gather(t, d), x, grad=True)
            both(f"reduce_scatter(dim={d})", lambda n, t, d=d: n.reduce_scatter(t, d), torch.randn([2 * world, 2 * world, 2 * world], generator=g), grad=True)
            both(f"spatial_split(dim={d})", lambda n, t, d=d: n.spatial_split(t, d), torch.randn([2 * world, 2 * world, 2 * world], generator=g), grad=True)
        both("allreduce_forward", lambda n, t: n.allreduce_forward(t), x, grad=True)
        both("allreduce_backward", lambda n, t: n.allreduce_backward(t), x, grad=True)
        both("zero_gather", lambda n, t: n.zero_gather(t), torch.randn([5, 3], generator=g), grad=True)
        both("zero_scatter", lambda n, t: n.zero_scatter(t, n.simple_split)[0], torch.randn([7, 3], generator=g))
        # TutelDistributedOptimizer (tutel/net.py:15-60): three SGD steps on a shared parameter of awkward size + an "expert" one
        finals = []
        for net in (rnet, mnet):
            torch.manual_seed(11)
            shared, expert = torch.nn.Parameter(torch.randn(7, 3)), torch.nn.Parameter(torch.randn(4, 2) + rank)
            expert._tutel_expert = True
            opt = net.TutelDistributedOptimizer([shared, expert], average_shared=True).warp_local(torch.optim.SGD, lr=0.1, momentum=0.5)
            for step in range(3):
                opt.zero_grad()
                ((shared * (rank + 1 + step)).sum() ** 2 + (expert ** 2).sum()).backward()
                opt.step()
            finals.append((shared.data.clone(), expert.data.clone()))
        if not all(torch.equal(a, b) for a, b in zip(*finals)):
            bad.append("TutelDistributedOptimizer: parameters after 3 steps differ")
        # group construction: same partitions
        for gc in (1, 2, -2):
            try:
                a, b = rnet.create_groups_from_world(group_count=gc), mnet.create_groups_from_world(group_count=gc)
                for attr in ("global_size", "global_rank", "group_count", "data_rank", "model_rank", "is_distributed"):
                    if getattr(a, attr, None) != getattr(b, attr, None):
                        bad.append(f"create_groups_from_world({gc}).{attr}: {getattr(a, attr, None)} vs {getattr(b, attr, None)}")
            except Exception as ex:   # noqa: BLE001
                skipped.append(f"create_groups_from_world({gc}): {type(ex).__name__}")
        q.put((rank, bad, skipped, True, None))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, [traceback.format_exc()], [], True, None))


def check_net_against_reference():
    import torch.multiprocessing as mp
    W = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_net_worker, args=(r, W, port, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = [q.get(timeout=300) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    n = 0
    for rank, bad, skipped, _, _ in sorted(res):
        for b in bad:
            print(f"MISMATCH (rank {rank}):", b)
            n += 1
        if rank == 0:
            for s_ in skipped:
                print("not compared:", s_)
    print("tutel.net, product vs reference over gloo: %d mismatches" % n)
    return n


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, case, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
        # the reference package is also called `tutel`: it must win over the repo's alias package
        sys.path = [REF, os.path.join(ROOT, "oracle", "_ref")] + [p for p in sys.path if os.path.abspath(p or ".") != ROOT] + [ROOT]
        import logging
        import torch
        import torch.distributed as dist
        logging.disable(logging.CRITICAL)
        import tutel
        assert os.path.abspath(tutel.__file__).startswith(REF), tutel.__file__
        from tutel import moe as ref_moe
        from oracle import moe_oracle as O
        dist.init_process_group("gloo", rank=rank, world_size=world)
        name, W, T, M, H, E_loc, k, dts, fp32_gate, tokens, cf = case
        dtype = getattr(torch, dts)
        E = E_loc * W
        n = (tokens or [T] * W)[rank]
        x = O.make_problem(T, M, H, E, dtype=dtype, seed=100 + rank)[0][:n]
        _, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=dtype, seed=7)
        sl = slice(rank * E_loc, (rank + 1) * E_loc)
        old = torch.get_default_dtype()
        torch.set_default_dtype(dtype)
        layer = ref_moe.moe_layer(gate_type={"type": "top", "k": k, "fp32_gate": fp32_gate, "capacity_factor": cf},
                                  experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                           "activation_fn": lambda t: torch.nn.functional.relu(t)}, model_dim=M)
        torch.set_default_dtype(old)
        with torch.no_grad():
            layer.gates[0].wg.weight.copy_(wg.to(layer.gates[0].wg.weight.dtype))
            layer.experts.batched_fc1_w.copy_(w1[sl]); layer.experts.batched_fc1_bias.copy_(b1[sl])
            layer.experts.batched_fc2_w.copy_(w2[sl]); layer.experts.batched_fc2_bias.copy_(b2[sl])
        layer.eval()
        got = []
        layer.experts.register_forward_pre_hook(lambda m, args: got.append(args[0].detach().clone()))
        with torch.no_grad():
            y = layer(x, inequivalent_tokens=tokens is not None)
        q.put((rank, y, got[0], layer.dispatch_count.to(torch.int32), float(y.l_aux)))
        dist.barrier()
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc(), None, None, None))


def run_reference(case):
    import torch.multiprocessing as mp
    W = case[1]
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, W, port, case, q)) for r in range(W)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=300) for _ in procs], key=lambda t: t[0])
    for p in procs:
        p.join(timeout=60)
    for r in res:
        assert r[2] is not None, r[1]
    return res


def oracle_forward(case):
    import torch
    sys.path.insert(0, ROOT)
    from oracle import moe_oracle as O
    name, W, T, M, H, E_loc, k, dts, fp32_gate, tokens, cf = case
    dtype = getattr(torch, dts)
    E = E_loc * W
    xs = [O.make_problem(T, M, H, E, dtype=dtype, seed=100 + r)[0][:(tokens or [T] * W)[r]] for r in range(W)]
    _, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=dtype, seed=7)
    parts = lambda t: [t[r * E_loc:(r + 1) * E_loc] for r in range(W)]
    return O.moe_forward_ep(xs, wg, parts(w1), parts(b1), parts(w2), parts(b2), top_k=k, capacity_factor=cf, fp32_gate=fp32_gate,
                            inequivalent_tokens=tokens is not None, return_expert_inputs=True)


def _np(t):
    import torch
    t = t.detach()
    return t.view(torch.int16).numpy() if t.dtype == torch.bfloat16 else t.numpy()


def main():
    import numpy as np
    import torch
    ap = argparse.ArgumentParser()
    ap.add_argument("--check", action="store_true")
    args = ap.parse_args()
    bad = 0
    for case in CASES:
        res = run_reference(case)
        if args.check:
            ys, crits, recvs = oracle_forward(case)
            for rank, y, recv, cnt, l_aux in res:
                ok = torch.equal(y, ys[rank]) and torch.equal(recv.reshape(recvs[rank].shape), recvs[rank]) and torch.equal(cnt, crits[rank][5])
                if not ok:
                    bad += 1
                    print("MISMATCH:", case[0], "rank", rank, float((y.double() - ys[rank].double()).abs().max()) if y.shape == ys[rank].shape else (y.shape, ys[rank].shape))
            continue
        out = {"meta": np.array([case[1], case[2], case[3], case[4], case[5], case[6], int(case[8])], dtype=np.int64), "dtype": np.array([case[7]]),
               "tokens": np.array(case[9] or [case[2]] * case[1], dtype=np.int64), "inequivalent": np.array([int(case[9] is not None)]),
               "cf": np.array([case[10]])}
        for rank, y, recv, cnt, l_aux in res:
            out[f"y_{rank}"], out[f"recv_{rank}"], out[f"count_{rank}"], out[f"l_aux_{rank}"] = _np(y), _np(recv), _np(cnt), np.array([l_aux])
        np.savez_compressed(os.path.join(HERE, f"ep_{case[0]}.npz"), **out)
        print("wrote", case[0], [tuple(r[1].shape) for r in res], [tuple(r[2].shape) for r in res])
    if args.check:
        print("oracle-vs-reference (expert parallel, W > 1): %d mismatches" % bad)
        bad += check_net_against_reference()
        sys.exit(1 if bad else 0)
    out = {"meta": np.array(SHARDED[1:], dtype=np.int64)}
    for rank, outs, _, _, _ in run_sharded_reference():
        for ptype, y in outs.items():
            out[f"y_{ptype.replace(':', '')}_{rank}"] = y
    np.savez_compressed(os.path.join(HERE, f"{SHARDED[0]}.npz"), **out)
    print("wrote", SHARDED[0], sorted(k for k in out if k != "meta"))


if __name__ == "__main__":
    main()
