"""Checkpoint re-sharding for expert-parallel MoE layers (reference: tutel/checkpoint/{gather,scatter}.py).

    python -m tutel.checkpoint.gather  --input_size=8 --inputs=ckpt/{rank}-of-{size}.pt --output=full.pt
    python -m tutel.checkpoint.scatter --output_size=4 --input=full.pt --outputs=new/{rank}-of-{size}.pt
"""
from .reshard import find_moe_layers, gather, merge_expert_shards, scatter, split_expert_param  # noqa: F401
