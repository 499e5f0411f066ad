"""Seeded fuzz of the expert-parallel forward between rank processes that share the one GPU (the machinery of test_ep_ranks_one_gpu.py):
random (tokens per rank, model / hidden dims, local experts, k, capacity rule, overlap degree, dtype, gate normalisation, score placement)
per case, a new layer per case over ONE process group and ONE communicator -- workspaces, exchange buffers and IPC segments of every shape
come and go -- each rank's output against the oracle's in-process simulation of the W ranks (oracle.moe_forward_ep) and its dispatch_count
element for element.  (Which pipeline a shape takes -- one-call, after-routing native, Python-orchestrated overlap, generic -- is the
planner's business, asserted by the dedicated tests; here every one of them must give the reference's answer.)  40 two-rank cases per transport in the default run; 300 two-rank and 150 three- / four-rank cases with --runslow."""
import os
import random

import pytest
import torch

from test_ep_ranks_one_gpu import _run_ranks, _set_transport

pytestmark = pytest.mark.gpu


def _fuzz_worker(rank, world, port, native, n_cases, seed, q):
    try:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0")
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import torch.distributed as dist
        from oracle import moe_oracle as O
        from tutel import moe
        from tutel_amd.impls import ep_native
        _set_transport(ep_native, native)
        dist.init_process_group("gloo", rank=rank, world_size=world)
        torch.cuda.set_device(0)
        rnd = random.Random(seed)   # the same draw on every rank
        bad = []
        for case in range(n_cases):
            E_loc = rnd.choice([1, 1, 2, 2, 3, 4, 8])
            degree = rnd.choice([1, 1, 2, 4])
            T = rnd.choice([64, 200, 512, 1000, 2048])
            M, H = rnd.choice([128, 128, 256, 512]), rnd.choice([128, 192, 320, 512])
            E = E_loc * world
            k = min(E, rnd.choice([1, 2, 2, 3]))
            cf = rnd.choice([1.0, 1.0, 1.0, 0.5, 2.0, 0.0, -1.0])
            dtype = rnd.choice([torch.bfloat16, torch.bfloat16, torch.float16])
            norm, post = rnd.random() < 0.7, rnd.random() < 0.7
            uneq = rnd.random() < 0.2
            Ts = [max(1, T - rnd.randrange(0, T // 2)) if (uneq and r > 0) else T for r in range(world)]
            use_2dh = case % 5 == 4      # the two-phase (intra-node, inter-node) all-to-all of BASELINE configs[4]: same values, another route
            tag = (f"case {case}: W={world} T={Ts} M={M} H={H} E_loc={E_loc} k={k} cf={cf} degree={degree} {dtype} norm={norm} post={post} "
                   f"use_2dh={use_2dh} transport={native}")
            xs = [O.make_problem(Ts[r], M, H, E, dtype=dtype, seed=seed * 1009 + case * 16 + r)[0] for r in range(world)]
            _, wg, w1, b1, w2, b2 = O.make_problem(8, M, H, E, dtype=dtype, seed=seed * 1009 + case * 16 + 15)
            sl = slice(rank * E_loc, (rank + 1) * E_loc)
            old = torch.get_default_dtype()
            torch.set_default_dtype(dtype)
            layer = moe.moe_layer(gate_type={"type": "top", "k": k, "fp32_gate": True, "capacity_factor": cf}, model_dim=M,
                                  experts={"type": "ffn", "num_experts_per_device": E_loc, "hidden_size_per_expert": H,
                                           "activation_fn": lambda t: torch.nn.functional.relu(t)},
                                  a2a_ffn_overlap_degree=degree, normalize_gate=norm, is_postscore=post, use_2dh=use_2dh)
            torch.set_default_dtype(old)
            with torch.no_grad():
                layer.gates[0].wg.weight.copy_(wg.float())
                layer.experts.batched_fc1_w.copy_(w1[sl]); layer.experts.batched_fc1_bias.copy_(b1[sl])
                layer.experts.batched_fc2_w.copy_(w2[sl]); layer.experts.batched_fc2_bias.copy_(b2[sl])
            layer = layer.cuda().eval()
            with torch.no_grad():
                y = layer(xs[rank].cuda(), inequivalent_tokens=uneq)
                y2 = layer(xs[rank].cuda(), inequivalent_tokens=uneq)   # cached workspaces / buffers: the same bits again
            torch.cuda.synchronize()
            box = [None]
            if rank == 0:
                sh = lambda t: [t[r * E_loc:(r + 1) * E_loc] for r in range(world)]
                box[0] = O.moe_forward_ep(xs, wg, sh(w1), sh(b1), sh(w2), sh(b2), top_k=k, capacity_factor=cf, fp32_gate=True,
                                          normalize_gate=norm, is_postscore=post, alignment=degree, accum_fp32=True, inequivalent_tokens=uneq)
            dist.broadcast_object_list(box, src=0)
            want, crits = box[0]
            err = (y.cpu().double() - want[rank].double()).abs()
            scale = float(want[rank].double().abs().max())
            tol = (2 ** -7 if dtype == torch.bfloat16 else 2 ** -10) * want[rank].double().abs() + max(2e-3, 2 ** -8 * scale)
            why = []
            if int((err > tol).sum()):
                why.append(f"{int((err > tol).sum())} elements over the bar (max err {float(err.max()):.3e}, |y|max {scale:.3f})")
            if not torch.equal(layer.dispatch_count.cpu(), crits[rank][5]):
                why.append("dispatch_count")
            if not torch.equal(y, y2):
                why.append("second forward of the same batch differs")
            if why:
                bad.append(f"rank {rank} {tag} :: " + "; ".join(why))
            del layer, y, y2
        q.put((rank, not bad, "\n".join(bad[:10]) or f"{n_cases} cases equal", []))
        dist.destroy_process_group()
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, False, traceback.format_exc(), []))


@pytest.mark.parametrize("world,n_cases", [(2, 40), pytest.param(2, 300, marks=pytest.mark.slow), pytest.param(3, 150, marks=pytest.mark.slow),
                                           pytest.param(4, 150, marks=pytest.mark.slow)])
@pytest.mark.parametrize("native", [False, True, "ipc"], ids=["python-orchestrated", "native-hosted-exchange", "native-ipc-peer-stores"])
def test_expert_parallel_fuzz_ranks_sharing_one_gpu(native, world, n_cases):
    _run_ranks(_fuzz_worker, world, (native, n_cases, 9090 + world), timeout=300 + 6 * n_cases)
