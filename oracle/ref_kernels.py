"""
oracle/ref_kernels.py -- TEST INFRASTRUCTURE ONLY (same rules as oracle/moe_oracle.py: tests/, smoke() and bench.py's
`cpu_baseline` leg may use it, nothing under tutel_amd/ does).

The reference's CPU path driven through the REFERENCE'S OWN compiled kernels: oracle/_ref/tutel_custom_kernel.so is
/root/reference/tutel/custom/custom_kernel.cpp built where it lies (oracle/Makefile, CPU configuration), and the two functions
below call its `invoke_cpu_fp32` (custom_kernel.cpp:280-323, bound at :758) with exactly the tensors, `extra` list and kernel
types that the reference's Python passes (tutel/impls/fast_dispatch.py:16-29 GatingEncoder.forward, :52-66 GatingDecoder.forward,
tutel/impls/jit_compiler.py:43-52 generate_cpu_kernel).  Everything around them -- softmax, top-k, the location cumsum, the expert
matmuls -- is ATen on the reference's CPU path as well, and is taken from moe_oracle.py, which restates those calls.

Why it exists: the reference's Python package cannot travel to the GPU box (no /root/reference there), its compiled kernels can
(oracle/_ref/ ships with the snapshot like any built .so).  bench.py's `cpu_baseline` therefore times THIS on the GPU box's host
cores (`kind: "reference"`: the reference's own scatter / gather loops + the ATen calls its Python makes) and the plain-C port
beside it.  tests/test_oracle_vs_reference.py::test_reference_kernels_equal_the_port compares the two bit for bit.
"""
import importlib.util
import os

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
_PATH = os.path.join(_HERE, "_ref", "tutel_custom_kernel.so")
_MOD = None


def available():
    return os.path.exists(_PATH)


def module():
    """the reference's extension module, loaded from oracle/_ref/ under the name its PyInit carries (tutel_custom_kernel)"""
    global _MOD
    if _MOD is None:
        if not available():
            raise FileNotFoundError(f"{_PATH}: build it with `make -C oracle` where /root/reference exists")
        spec = importlib.util.spec_from_file_location("tutel_custom_kernel", _PATH)
        _MOD = importlib.util.module_from_spec(spec)
        spec.loader.exec_module(_MOD)
    return _MOD


def _args(crit, j):
    _, idx_list, loc_list, _, _, _ = crit
    # fast_dispatch.py:102-103: indices_ = x.to(int32).view(-1), locations_ = x.to(int32)
    return idx_list[j].to(torch.int32).view(-1).contiguous(), loc_list[j].to(torch.int32).contiguous()


def fast_encode(x, crit, is_postscore=True):
    """[T, M] -> [E, C, M] through the reference's kernel_type 0 (custom_kernel.cpp:293-300).  fast_dispatch.py:123-127 (encode),
    :16-29: zeros([E * C, M]) in the dispatch dtype (fp32 on a HIP build, :94-96), one kernel call per top-k choice with the
    [S, 2] ones helper (:116-120) when the gates are applied at decode."""
    E, idx_list, _, gates, C, _ = crit
    T, M = x.shape
    xin = x.to(torch.float32).contiguous()
    out = torch.zeros([E * C, M], dtype=torch.float32)
    ones = torch.ones([T, 2], dtype=torch.float32)
    for j in range(len(idx_list)):
        g = ones if is_postscore else gates[j].to(torch.float32).contiguous()
        i32, l32 = _args(crit, j)
        module().invoke_cpu_fp32([g, i32, l32, xin, out], [T, M, C], 0)
    return out.to(x.dtype).view(E, C, M)


def fast_decode(y, crit, is_postscore=True):
    """[E, C, M] -> [T, M] through the reference's kernel_type 1 (custom_kernel.cpp:301-312).  fast_dispatch.py:129-133 (decode),
    :52-66: one empty([T, M]) per choice, `last_result + single_output` left to right, one cast at the end."""
    E, idx_list, _, gates, C, _ = crit
    M = y.shape[-1]
    T = idx_list[0].numel()
    buf = y.reshape(E * C, M).to(torch.float32).contiguous()
    ones = torch.ones([T, 2], dtype=torch.float32)
    acc = None
    for j in range(len(idx_list)):
        g = gates[j].to(torch.float32).contiguous() if is_postscore else ones
        i32, l32 = _args(crit, j)
        tmp = torch.empty([T, M], dtype=torch.float32)
        module().invoke_cpu_fp32([g, i32, l32, tmp, buf], [T, M, C], 1)
        acc = tmp if acc is None else acc + tmp
    return acc.to(y.dtype)


def moe_forward(x, wg, w1, b1, w2, b2, **kw):
    """moe_oracle.moe_forward with the reference's compiled kernels as fast_encode / fast_decode."""
    from . import moe_oracle as O
    return O.moe_forward(x, wg, w1, b1, w2, b2, encode_fn=fast_encode, decode_fn=fast_decode, **kw)
