"""ctypes binding of libtutel_amd.so (C ABI: include/tutel_amd.h).

The HIP library IS the product's compute path.  There is no CPU / eager fallback: if the
library is missing or a call fails, this module raises -- loudly -- instead of computing the
result some other way.
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libtutel_amd.so")
CSRC_DIR = os.path.join(_HERE, "csrc")

F32, F16, BF16, F64 = 0, 1, 2, 3
ACT_NONE, ACT_RELU, ACT_GELU, ACT_SILU = 0, 1, 2, 3
OPT_GEMM_IMPL, OPT_GEMM_TILE, OPT_DECODE, OPT_EP_STAGE_GRID, OPT_GEMM_PERSIST, OPT_EP_STREAMS, OPT_EP_CANARY, OPT_GEMM_SPLITK, OPT_GEMM_GATHER, OPT_FUSED_LOCATION, OPT_GEMM_STORE, OPT_TIE_RULE, OPT_FFN_FUSED = 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12

_vp, _i, _i64, _sz = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_size_t

# name -> (restype, argtypes); must list every symbol include/tutel_amd.h declares.
SIGNATURES = {
    "tutel_amd_abi_version": (_i, []),
    "tutel_amd_target_arch": (ctypes.c_char_p, []),
    "tutel_amd_last_error": (ctypes.c_char_p, []),
    "tutel_amd_routing_workspace_bytes": (_sz, [_i, _i, _i]),
    "tutel_amd_gate_topk": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp]),
    "tutel_amd_gate_proj_splits": (_i, [_i, _i, _i, _i]),
    "tutel_amd_gate_proj": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _sz, _vp]),
    "tutel_amd_gate_topk_partials": (_i, [_vp, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _sz, _vp, _i, _vp]),
    "tutel_amd_cache_warm": (_i, [_vp, _sz, _i, _sz, _i, _vp, _vp]),
    "tutel_amd_compute_location": (_i, [_vp, _i, _i, _i, _i, _vp, _sz, _vp, _vp, _vp, _vp, _i, _i, _vp, _i, _vp]),
    "tutel_amd_slot_map": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "tutel_amd_cumsum_sub_one": (_i, [_vp, _vp, _i, _i, _vp]),
    "tutel_amd_fast_encode": (_i, [_vp, _i, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "tutel_amd_fast_decode": (_i, [_vp, _i, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _vp]),
    "tutel_amd_gate_grad": (_i, [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _vp]),
    "tutel_amd_expert_gemm": (_i, [_vp, _i64, _i64, _i, _i, _vp, _i, _i64, _i, _vp, _i64, _vp, _i64,
                                   _i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "tutel_amd_expert_gemm_gather": (_i, [_vp, _i, _vp, _i, _vp, _vp, _i, _i64, _i, _vp, _i64, _vp, _i64, _i,
                                          _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "tutel_amd_expert_ffn": (_i, [_vp, _i64, _i, _vp, _i, _vp, _vp, _i64, _i, _vp, _i64, _vp, _i64, _i, _vp, _i64, _i, _vp, _i64, _vp, _i64, _i,
                                  _i, _i, _i, _i, _i, _i, _i, _vp]),
    "tutel_amd_expert_gemm_glu": (_i, [_vp, _i64, _i64, _i, _i, _vp, _i, _i64, _i, _vp, _i64, _vp, _vp, _i64,
                                       _i64, _i, _i, _i, _i, _i, _i, _i, _i, _vp, _i, _vp]),
    "tutel_amd_set_option": (_i, [_i, _i]),
    "tutel_amd_probe_tr16": (_i, [_vp, _vp]),
}



class EpPlan(ctypes.Structure):
    """tutel_amd_ep_plan_t"""
    _fields_ = [(n, _i) for n in ("sliced", "experts_per_stage", "chunk", "rows", "gemm_rows")]


class EpArgs(ctypes.Structure):
    """tutel_amd_ep_args_t (include/tutel_amd.h): field order and types must match the header"""
    _fields_ = ([(n, _i) for n in ("T", "M", "H", "M_out", "num_experts", "world", "k", "capacity", "degree", "allow_sliced",
                                   "dtype", "gate_dtype", "act", "is_postscore", "w2_kmajor", "fuse_encode")] +
                [(n, _vp) for n in ("x", "slot_map", "idx", "loc", "gates", "w1", "b1", "w2", "b2",
                                    "enc", "recv", "hid", "send", "back", "zero_row", "row_counts")] +
                [("row_align", _i), ("y", _vp), ("peer_seg", _vp)])


class MoeArgs(ctypes.Structure):
    """tutel_amd_moe_args_t"""
    _fields_ = [("ep", EpArgs), ("logits", _vp), ("logits_dtype", _i), ("normalize_gate", _i), ("ws", _vp), ("ws_bytes", _sz),
                ("dispatch_count", _vp), ("stats", _vp), ("l_aux", _vp),
                ("capacity_limit", _i), ("alignment", _i), ("max_capacity", _i), ("capacity_out", ctypes.POINTER(_i)),
                ("gate_w", _vp), ("gate_partials", _vp), ("gate_partial_bytes", _sz), ("logits_out", _vp),
                ("fl_ws", _vp), ("fl_ws_bytes", _sz)]


SIGNATURES.update({
    "tutel_amd_moe_forward": (_i, [_vp, ctypes.POINTER(MoeArgs), _vp]),
    "tutel_amd_ep_load_rccl": (_i, [ctypes.c_char_p]),
    "tutel_amd_ep_unique_id": (_i, [_vp, _sz]),
    "tutel_amd_ep_comm_create": (_i, [_vp, _sz, _i, _i, ctypes.POINTER(_vp)]),
    "tutel_amd_ep_comm_destroy": (_i, [_vp]),
    "tutel_amd_ep_comm_create_hosted": (_i, [_i, _i, _vp, _vp, ctypes.POINTER(_vp)]),
    "tutel_amd_ep_comm_info": (_i, [_vp, ctypes.POINTER(_i), ctypes.POINTER(_i)]),
    "tutel_amd_ep_all_to_all": (_i, [_vp, _vp, _vp, _sz, _vp]),
    "tutel_amd_ep_comm_set_hosted_v": (_i, [_vp, _vp]),
    "tutel_amd_ep_all_to_all_v": (_i, [_vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint64), ctypes.POINTER(ctypes.c_uint64), _vp]),
    "tutel_amd_ep_all_gather_v": (_i, [_vp, _vp, _vp, ctypes.POINTER(ctypes.c_uint64), _vp]),
    "tutel_amd_ep_plan": (_i, [_i, _i, _i, _i, _i, ctypes.POINTER(EpPlan)]),
    "tutel_amd_ep_forward": (_i, [_vp, ctypes.POINTER(EpArgs), _vp]),
    "tutel_amd_ep_segment_alloc": (_i, [_sz, _i, ctypes.POINTER(_vp), _vp, _sz]),
    "tutel_amd_ep_segment_open": (_i, [_vp, _i, _i, _vp, _sz]),
    "tutel_amd_ep_segment_ptr": (_vp, [_vp, _i]),
    "tutel_amd_ep_segment_read": (_i, [_vp, _sz, _vp, _sz, _vp]),
    "tutel_amd_ep_segment_free": (_i, [_vp]),
    "tutel_amd_ep_flag_bytes": (_sz, []),
    "tutel_amd_ep_comm_create_ipc": (_i, [_i, _i, ctypes.POINTER(_vp)]),
    "tutel_amd_ep_comm_attach_ipc": (_i, [_vp, _vp, _i]),
    "tutel_amd_ep_comm_has_ipc": (_i, [_vp]),
    "tutel_amd_ep_ipc_set_timeout": (_i, [_vp, _i]),
    "tutel_amd_ep_ipc_status": (_i, [_vp]),
    "tutel_amd_ep_ipc_exchange": (_i, [_vp, _vp, _vp, _sz, _sz, _vp]),
    "tutel_amd_ep_ipc_selfcheck": (_i, [_vp, _vp, _sz, _i, _i, _i, _vp, _vp]),
    "tutel_amd_mark": (_i, [_vp]),
    "tutel_amd_marks_reserve": (_i, [_i]),
    "tutel_amd_marks_report": (_i, [ctypes.POINTER(ctypes.c_double), _i]),
    "tutel_amd_stage_timing": (_i, [_i]),
    "tutel_amd_stage_report": (_i, [ctypes.POINTER(ctypes.c_double), ctypes.POINTER(_i), _i]),
    "tutel_amd_range_push": (_i, [ctypes.c_char_p]),
    "tutel_amd_range_pop": (_i, []),
})
EXCHANGE_FN = ctypes.CFUNCTYPE(_i, _vp, _vp, _vp, _sz, _i)
_u64p = ctypes.POINTER(ctypes.c_uint64)
EXCHANGE_V_FN = ctypes.CFUNCTYPE(_i, _vp, _vp, _vp, _u64p, _u64p, _u64p, _i)
EP_ID_BYTES = 128
IPC_HANDLE_BYTES = 64
EAGAIN, ENOTSUP = 1000, 1001
STAGES = ("gate_topk", "location", "fast_encode", "expert_fc1", "expert_fc2", "fast_decode", "all_to_all_dispatch", "all_to_all_combine", "other",
          "gate_projection")

_lib = None


class TutelAmdError(RuntimeError):
    pass


def build(force=False):
    """Compile libtutel_amd.so for gfx950 with hipcc (cross-compiles without a GPU)."""
    if force or not os.path.exists(LIB_PATH):
        subprocess.check_call(["make", "-C", CSRC_DIR, "-j8"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    """The loaded library.  Raises TutelAmdError when it is not there -- never falls back."""
    global _lib
    if _lib is None:
        # One HIP runtime per process: PyTorch's wheel bundles its own libamdhip64 / libhsa-runtime64 and
        # loads them by path.  If libtutel_amd.so came first it would pull /opt/rocm's copies in, torch
        # would then add its own, and whichever runtime initialises second finds "no ROCm-capable
        # device".  Importing torch first makes the loader satisfy our NEEDED libamdhip64.so.7 with the
        # copy torch already mapped, so kernels launched here run on torch's streams and allocations.
        import torch  # noqa: F401
        if not os.path.exists(LIB_PATH):
            raise TutelAmdError(
                f"tutel_amd: HIP library {LIB_PATH} is missing. Build it with "
                f"`python -c 'import __graft_entry__ as g; g.build()'` or `make -C tutel_amd/csrc` "
                f"(hipcc --offload-arch=gfx950). There is no CPU fallback for the MoE hot path.")
        try:
            handle = ctypes.CDLL(LIB_PATH)
        except OSError as ex:  # e.g. libamdhip64 missing
            raise TutelAmdError(f"tutel_amd: cannot load {LIB_PATH}: {ex}") from ex
        for name, (res, args) in SIGNATURES.items():
            try:
                fn = getattr(handle, name)
            except AttributeError as ex:
                raise TutelAmdError(f"tutel_amd: {LIB_PATH} does not export `{name}` (stale build?)") from ex
            fn.restype, fn.argtypes = res, args
        if handle.tutel_amd_abi_version() != 1:
            raise TutelAmdError("tutel_amd: ABI version mismatch between python and libtutel_amd.so")
        _lib = handle
    return _lib


def check(status, what):
    if status != 0:
        msg = lib().tutel_amd_last_error().decode("utf-8", "replace")
        raise TutelAmdError(f"{what} failed (status {status}): {msg}")
