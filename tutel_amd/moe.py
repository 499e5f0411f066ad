"""tutel.moe facade: low-level ops + the layer (reference: tutel/moe.py)."""
from .jit_kernels.gating import fast_cumsum_sub_one
from .impls.fast_dispatch import fast_dispatcher, extract_critical, fast_encode, fast_decode
from .impls.moe_layer import moe_layer

top_k_routing = extract_critical
