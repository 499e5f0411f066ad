#!/usr/bin/env python3
"""Sweep driver of BASELINE configs[4] (reference: tutel/examples/helloworld_switch.py:84-88): one MoE layer, every step
takes the next (adaptive_r, a2a_ffn_overlap_degree) pair -- r over the layer's valid_rs, overlap degree 1..8 -- and prints
its step time, so a run of len(valid_rs) * 8 steps times every combination.

    python -m torch.distributed.run --nproc-per-node 8 --master-addr 127.0.0.1 -m tutel_amd.examples.helloworld_switch \
           --eval --dtype=float16 --batch_size=16 --num_tokens=512 --model_dim=4096 --hidden_size=4096 \
           --num_local_experts=16 --use_2dh --num_steps=32
"""
from .helloworld import main

if __name__ == "__main__":
    main(switch=True)
