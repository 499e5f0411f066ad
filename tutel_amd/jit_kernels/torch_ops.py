"""`torch.ops.tutel_ops.cumsum` and `torch.ops.tutel_ops.sparse_bmm_infer` -- the two operator names the reference registers from
its extension (TORCH_LIBRARY(tutel_ops), custom_kernel.cpp:822-894) and that user / custom expert code calls the way
experts/ffn.py:70-81 does -- defined over the C ABI of libtutel_amd.so (tutel_amd_cumsum_sub_one, tutel_amd_expert_gemm with
device-side row counts).

    cumsum(Tensor x) -> Tensor                       x [T, E] (any integer dtype) -> int32 per-column inclusive cumsum - 1
    sparse_bmm_infer(Tensor x, Tensor w, Tensor sparse_groups_device, bool w_transpose, int sparse_size) -> Tensor
                                                     y[e, :g_e * s] = x[e, :g_e * s] @ (w[e].T if w_transpose else w[e]);
                                                     the rows past g_e * s are left unwritten, as upstream's torch.empty leaves them

Registered once per process at import (`tutel.jit_kernels.gating`, like upstream's torch.ops.load_library there).  If another
library already owns the `tutel_ops` namespace (upstream Tutel's GPU extension in the same process) nothing is registered."""
import torch

from .. import ops

_lib_def = None


def _cumsum(x):
    if x.dim() != 2:
        raise RuntimeError("tutel_ops::cumsum: expected a 2-D tensor, got %s" % (list(x.shape),))
    return ops.cumsum_sub_one(x.to(torch.int32).contiguous())


def _sparse_bmm_infer(x, w, sparse_groups_device, w_transpose, sparse_size):
    if x.dim() != 3 or w.dim() != 3 or x.size(0) != w.size(0):
        raise RuntimeError("tutel_ops::sparse_bmm_infer: expected x [E, R, K] and w [E, ., .]")
    E, R, K = x.shape
    N = w.size(1) if w_transpose else w.size(2)
    if (w.size(2) if w_transpose else w.size(1)) != K:
        raise RuntimeError("tutel_ops::sparse_bmm_infer: inner dimensions do not match")
    rows = (sparse_groups_device.to(device=x.device, dtype=torch.int32) * int(sparse_size)).clamp_(max=R).contiguous()
    if x.dtype in (torch.bfloat16, torch.float16) and w.dtype == x.dtype and K % 64 == 0 and N % 8 == 0 and x.is_cuda:
        # one launch for every expert, the row counts stay on the device (upstream: .cpu() sync + one matmul per expert)
        return ops.expert_gemm(x.contiguous(), w.contiguous(), None, bool(w_transpose), row_counts=rows, row_align=1)
    # dtypes / shapes the MFMA kernel does not take: upstream's own op sequence (custom_kernel.cpp:874-889)
    y = torch.empty([E, R, N], dtype=x.dtype, device=x.device)
    for e, n in enumerate(rows.cpu().tolist()):
        if n > 0:
            torch.matmul(x[e, :n], w[e].t() if w_transpose else w[e], out=y[e, :n])
    return y


def register():
    """idempotent; returns True when torch.ops.tutel_ops.{cumsum, sparse_bmm_infer} resolve to this library"""
    global _lib_def
    if _lib_def is not None:
        return True
    try:
        lib = torch.library.Library("tutel_ops", "DEF")
        lib.define("cumsum(Tensor x) -> Tensor")
        lib.define("sparse_bmm_infer(Tensor x, Tensor w, Tensor sparse_groups_device, bool w_transpose, int sparse_size) -> Tensor")
        lib.impl("cumsum", _cumsum, "CUDA")
        lib.impl("sparse_bmm_infer", _sparse_bmm_infer, "CUDA")
    except RuntimeError:   # the namespace is taken (upstream's extension is loaded): leave it alone
        return False
    _lib_def = lib
    return True
