"""fast_cumsum_sub_one (reference: tutel/jit_kernels/gating.py:19-24 over
torch.ops.tutel_ops.cumsum, custom_kernel.cpp:822-872) on the HIP library."""
from .. import ops
from . import torch_ops

has_extension = torch_ops.register()   # torch.ops.tutel_ops.cumsum / sparse_bmm_infer (custom_kernel.cpp:891-894)


def fast_cumsum_sub_one(data, dim=0):
    if data.dim() != 2 or dim != 0:
        raise Exception("Unimplemented fast_cumsum_sub_one() of data = %s and dim = %s" % (data.size(), dim))
    return ops.cumsum_sub_one(data)
