"""`tutel.moe` facade (reference: tutel/moe.py): the layer plus the low-level routing / dispatch ops.

    moe_layer                      impls/moe_layer.py    (MI355X forward path)
    top_k_routing|extract_critical impls/fast_dispatch.py -> tutel_amd_gate_topk + tutel_amd_compute_location
    fast_encode / fast_decode      impls/fast_dispatch.py -> tutel_amd_fast_encode / tutel_amd_fast_decode
    fast_dispatcher                impls/fast_dispatch.py (TutelMoeFastDispatcher)
    fast_cumsum_sub_one            jit_kernels/gating.py  -> tutel_amd_cumsum_sub_one
"""
from .impls import fast_dispatch as _dispatch
from .impls import moe_layer as _layer
from .jit_kernels import gating as _gating

moe_layer = _layer.MOELayer
extract_critical = top_k_routing = _dispatch.extract_critical
fast_encode, fast_decode = _dispatch.fast_encode, _dispatch.fast_decode
fast_dispatcher = _dispatch.TutelMoeFastDispatcher
fast_cumsum_sub_one = _gating.fast_cumsum_sub_one

__all__ = ["moe_layer", "top_k_routing", "extract_critical", "fast_encode", "fast_decode",
           "fast_dispatcher", "fast_cumsum_sub_one"]
