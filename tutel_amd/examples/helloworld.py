#!/usr/bin/env python3
"""helloworld driver with the reference's flag surface (tutel/examples/helloworld.py:17-42):
one MoE layer on synthetic tokens, training (fwd+bwd+SGD) or --eval forward, prints loss /
step_time / tflops per step and the average of the last 10 steps.

    python -m tutel_amd.examples.helloworld --eval --dtype=bfloat16 --num_local_experts=64 \
           --batch_size=16 --num_tokens=256                         # BASELINE configs[1]
    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 \
           -m tutel_amd.examples.helloworld --eval --num_local_experts=8 --a2a_ffn_overlap_degree=2
"""
import argparse
import os

import torch
import torch.nn.functional as F

from tutel import moe as tutel_moe
from tutel import net, system


def main(argv=None, switch=False, amp=False):
    ap = argparse.ArgumentParser()
    ap.add_argument("--local_rank", type=int, default=-1)
    ap.add_argument("--batch_size", type=int, default=16)
    ap.add_argument("--num_tokens", type=int, default=512)
    ap.add_argument("--model_dim", type=int, default=2048)
    ap.add_argument("--hidden_size", type=int, default=2048)
    ap.add_argument("--num_local_experts", type=int, default=2)
    ap.add_argument("--dtype", type=str, default="float32")
    ap.add_argument("--fp32_gate", default=False, action="store_true")
    ap.add_argument("--top", type=int, default=2)
    ap.add_argument("--l_aux_wt", type=float, default=0.0)
    ap.add_argument("--a2a_ffn_overlap_degree", type=int, default=1)
    ap.add_argument("--allreduce_degree", type=int, default=1)
    ap.add_argument("--num_steps", type=int, default=100)
    ap.add_argument("--parallel_type", type=str, default="adaptive:1")
    ap.add_argument("--device", type=str, default="cuda")
    ap.add_argument("--use_2dh", default=False, action="store_true")
    ap.add_argument("--eval", default=False, action="store_true")
    ap.add_argument("--capacity_factor", type=float, default=1.0)
    ap.add_argument("--megablocks_size", type=int, default=0)
    ap.add_argument("--checkpoint_path", type=str, default="")
    ap.add_argument("--use_tensorcore", default=False, action="store_true")  # accepted for CLI compatibility: no TF32 on gfx950
    ap.add_argument("--expert_type", type=str, default="ffn")
    ap.add_argument("--cap_factor", type=float, default=None, help="helloworld_switch.py's name for --capacity_factor")
    ap.add_argument("--switch", default=switch, action="store_true",
                    help="helloworld_switch.py: every step takes the next (adaptive_r, a2a_ffn_overlap_degree) of valid_rs x 1..8")
    ap.add_argument("--amp", default=amp, action="store_true", help="helloworld_amp.py: forward under torch.autocast")
    args = ap.parse_args(argv)
    if args.cap_factor is not None:
        args.capacity_factor = args.cap_factor

    env = system.init_data_model_parallel(backend="nccl" if args.device == "cuda" else "gloo")
    rank, world, dprint, device = env.global_rank, env.global_size, env.dist_print, env.local_device
    dtype = {"float32": torch.float32, "float64": torch.float64, "float16": torch.float16, "bfloat16": torch.bfloat16}[args.dtype]
    torch.set_default_dtype(dtype)

    layer = tutel_moe.moe_layer(
        gate_type={"type": "top", "k": args.top, "fp32_gate": args.fp32_gate, "capacity_factor": args.capacity_factor},
        experts={"type": args.expert_type, "num_experts_per_device": args.num_local_experts,
                 "hidden_size_per_expert": args.hidden_size, "activation_fn": lambda x: F.relu(x)},
        model_dim=args.model_dim, scan_expert_func=lambda n, p: setattr(p, "skip_allreduce", True),
        seeds=(1, rank + 1, 1), a2a_ffn_overlap_degree=args.a2a_ffn_overlap_degree,
        parallel_type=args.parallel_type, use_2dh=args.use_2dh).to(device)
    dprint(layer)
    checkpoint_path = ""
    if args.checkpoint_path:  # per-rank files, pattern with {rank} / {size} (helloworld.py:103-108)
        checkpoint_path = system.apply_rank_size_from_pattern(args.checkpoint_path, rank=rank, size=world)
        if os.path.exists(checkpoint_path):
            layer.load_state_dict(torch.load(checkpoint_path))
        else:
            print("Checkpoint not loaded: file `%s` is not found. Will train the model from start." % checkpoint_path)

    state = {"r_index": -1}

    def model(inp):
        if args.switch:  # the sweep of the reference's helloworld_switch.py:84-88: r over valid_rs, overlap degree 1..8
            rs = layer.valid_rs
            r, o = rs[(state["r_index"] // 8) % len(rs)], state["r_index"] % 8 + 1
            state["r_index"] += 1
            out = layer(inp, capacity_factor=args.capacity_factor, adaptive_r=r, a2a_ffn_overlap_degree=o)
        elif args.megablocks_size > 0:
            out = layer(inp, megablocks_size=args.megablocks_size)
        else:
            out = layer(inp)
        return F.log_softmax(torch.sum(out, dim=2), dim=1)

    if args.amp:  # helloworld_amp.py:76-79: the whole forward under autocast (fp32 master weights)
        plain_model = model

        def model(inp):  # noqa: F811
            with torch.autocast(device.type if device is not None else "cuda"):
                return plain_model(inp)

    opt = torch.optim.SGD(layer.parameters(), lr=1e-5)
    torch.manual_seed(0)
    x = torch.randn([args.batch_size, args.num_tokens, args.model_dim], dtype=torch.float32, device="cpu").to(dtype).to(device)
    y = torch.zeros(args.batch_size, dtype=torch.long, device=device)
    shared = [p for p in layer.parameters() if not hasattr(p, "skip_allreduce") and p.requires_grad]
    dprint("[Benchmark] world_size = %s, dtype = %s, model_dim = %s, hidden_size = %s, samples = %s, num_local_experts = %s, topK = %s, a2a_ffn_overlap_degree = %s, device = `%s`"
           % (world, args.dtype, args.model_dim, args.hidden_size, args.batch_size * args.num_tokens, args.num_local_experts, args.top, args.a2a_ffn_overlap_degree, device))

    avg = 0.0
    for i in range(args.num_steps):
        t0 = system.record_time()
        if not args.eval:
            opt.zero_grad()
            loss = F.nll_loss(model(x), y)
            if args.l_aux_wt:
                loss = loss + args.l_aux_wt * layer.l_aux
            loss.backward()
            if world > 1:
                for p in shared:
                    p.grad /= world
                    p.grad = net.simple_all_reduce(p.grad)
            opt.step()
        else:
            with torch.no_grad():
                loss = F.nll_loss(model(x), y)
        t1 = system.record_time()
        E = tutel_moe.moe_layer.global_expert_count(args.num_local_experts, group=system.get_local_session().model_group)
        tflops = (args.batch_size * args.num_tokens * args.model_dim * args.hidden_size) * 4 * (1 if args.eval else 3) * min(args.top, E) * 1e-12 / (t1 - t0)
        if args.switch:
            dprint("STEP-%s: loss = %.5f, step_time = %.6f sec, perf = %.2f tflops. (f = %.1f, r = %d, o = %d)"
                   % (i, float(loss.data), t1 - t0, tflops, args.capacity_factor, layer.adaptive_degree, layer.a2a_ffn_overlap_degree))
        else:
            dprint("STEP-%s: loss = %.5f, step_time = %.6f sec, perf = %.2f tflops." % (i, float(loss.data), t1 - t0, tflops))
        if i + 10 >= args.num_steps:
            avg += t1 - t0
    dprint("\n[Summary] Average synchronized step_time = %s sec." % (avg / 10))
    if checkpoint_path:
        torch.save(layer.state_dict(), checkpoint_path)


if __name__ == "__main__":
    main()
