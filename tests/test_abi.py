"""The C-ABI boundary without a GPU: the library builds/loads, exports exactly what
include/tutel_amd.h declares, and rejects bad arguments before enqueuing anything."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def L():
    from tutel_amd import _lib
    _lib.build()
    return _lib.lib()


def _declared():
    hdr = open(os.path.join(ROOT, "include", "tutel_amd.h")).read()
    hdr = re.sub(r"/\*.*?\*/", "", hdr, flags=re.S)
    return sorted(set(re.findall(r"\b(tutel_amd_\w+)\s*\(", hdr)))


def test_header_symbols_exported_and_bound(L):
    from tutel_amd import _lib
    names = _declared()
    assert len(names) >= 13
    assert set(names) == set(_lib.SIGNATURES), "python binding must cover exactly the header's entry points"
    raw = ctypes.CDLL(_lib.LIB_PATH)
    for n in names:
        assert hasattr(raw, n), f"{n} declared in include/tutel_amd.h but not exported"


def test_library_info(L):
    assert L.tutel_amd_abi_version() == 1
    assert L.tutel_amd_target_arch() == b"gfx950"
    assert L.tutel_amd_routing_workspace_bytes(4096, 64, 2) == 64 * (2 * 64 * 4 + 64 * 4)
    assert L.tutel_amd_routing_workspace_bytes(0, 64, 2) == 0


def test_argument_errors_are_reported_not_enqueued(L):
    from tutel_amd import _lib
    # unsupported dtype
    assert L.tutel_amd_fast_encode(None, 99, None, None, 0, 4, 8, 4, 0, 0, 0, 0, 1, None, None) != 0
    assert b"dtype" in L.tutel_amd_last_error()
    # k > E
    assert L.tutel_amd_gate_topk(None, 0, 0, 4, 2, 3, 1, None, None, None, None, 0, None, 0, None) != 0
    assert b"k" in L.tutel_amd_last_error()
    # GEMM: K not a multiple of 64, fp32 experts
    assert L.tutel_amd_expert_gemm(None, 0, 0, 1, 0, None, 1, 0, 0, None, 0, None, 0, 0, 1, 0, 1, 1, 8, 100, 2, 0, None, 1, None) != 0
    assert L.tutel_amd_expert_gemm(None, 0, 0, 1, 0, None, 1, 0, 0, None, 0, None, 0, 0, 1, 0, 1, 1, 8, 64, 0, 0, None, 1, None) != 0
    with pytest.raises(_lib.TutelAmdError):
        _lib.check(L.tutel_amd_fast_decode(None, 7, None, None, None, 0, 1, 1, 1, 1, 0, 0, 0, 1, None, None), "decode")
    # empty problems are a no-op success (nothing to launch)
    assert L.tutel_amd_fast_decode(None, 0, None, None, None, 0, 0, 8, 2, 4, 0, 0, 0, 1, None, None) == 0
    assert L.tutel_amd_fast_decode(None, 0, None, None, None, 0, 4, 8, 2, 6, 3, 4, 0, 1, None, None) != 0  # chunk must divide capacity
    assert L.tutel_amd_fast_decode(None, 0, None, None, None, 0, 4, 8, 2, 6, 8, 0, 3, 2, None, None) != 0  # slice must divide E_loc
    assert L.tutel_amd_expert_gemm(None, 0, 0, 1, 0, None, 1, 0, 0, None, 0, None, 0, 0, 1, 0, 0, 0, 8, 64, 2, 0, None, 1, None) == 0


def test_round3_entry_points_reject_bad_arguments(L):
    from tutel_amd import _lib
    # IPC transport: a communicator without it refuses, segments check their arguments
    assert L.tutel_amd_ep_ipc_exchange(None, None, None, 16, 0, None) != 0 and b"IPC" in L.tutel_amd_last_error()
    assert L.tutel_amd_ep_segment_open(None, 2, 0, None, 64) != 0
    assert L.tutel_amd_ep_comm_create_ipc(17, 0, None) != 0 and L.tutel_amd_ep_comm_has_ipc(None) == 0
    assert L.tutel_amd_ep_flag_bytes() >= 2 * 33 * 16 * 4
    # variable-size collectives need a communicator
    u64 = (ctypes.c_uint64 * 2)(8, 8)
    assert L.tutel_amd_ep_all_to_all_v(None, None, None, u64, u64, None) != 0 and b"communicator" in L.tutel_amd_last_error()
    assert L.tutel_amd_ep_all_gather_v(None, None, None, u64, None) != 0 and b"communicator" in L.tutel_amd_last_error()
    assert L.tutel_amd_ep_comm_set_hosted_v(None, None) != 0
    # limits named in the error text (INTEGRATION.md "Limits")
    assert L.tutel_amd_gate_topk(None, 0, 0, 4, 4097, 1, 1, None, None, None, None, 0, None, 0, None) != 0 and b"4096" in L.tutel_amd_last_error()
    assert L.tutel_amd_gate_topk(None, 0, 0, 4, 4096, 3, 1, None, None, None, None, 0, None, 0, None) != 0 and b"8192" in L.tutel_amd_last_error()
    # options: every key the header defines exists (and Python's copy of the numbering agrees with it), unknown keys are refused
    import re
    hdr = open(os.path.join(ROOT, "include", "tutel_amd.h")).read()
    keys = {m.group(1): int(m.group(2)) for m in re.finditer(r"#define TUTEL_OPT_(\w+) (\d+)", hdr)}
    count = keys.pop("COUNT")
    assert sorted(keys.values()) == list(range(count)) and keys["FFN_FUSED"] == _lib.OPT_FFN_FUSED == count - 1
    for name, key in keys.items():
        assert L.tutel_amd_set_option(key, -1) == 0, name
        assert getattr(_lib, "OPT_" + name) == key, name
    assert L.tutel_amd_set_option(count, 0) != 0 and L.tutel_amd_set_option(99, 0) != 0


def test_product_has_no_cpu_path():
    import torch
    from tutel_amd import _lib, ops
    with pytest.raises(_lib.TutelAmdError):
        ops.fast_decode(torch.zeros(4, 8), torch.zeros(1, 2, dtype=torch.int32), torch.zeros(1, 2, dtype=torch.int32), None, 2)
    # nothing under tutel_amd/ or tutel/ references the oracle
    for base in ("tutel_amd", "tutel"):
        for dp, _, files in os.walk(os.path.join(ROOT, base)):
            for f in files:
                if f.endswith((".py", ".hip", ".h")):
                    assert "oracle" not in open(os.path.join(dp, f)).read().lower(), os.path.join(dp, f)


def test_header_is_plain_c(tmp_path):
    """the drop-in boundary is a C ABI: include/tutel_amd.h must compile as C99 on its own (no C++, no HIP or torch headers)"""
    import subprocess
    src = tmp_path / "h.c"
    src.write_text('#include "tutel_amd.h"\nint main(void) { tutel_amd_ep_args_t a; tutel_amd_moe_args_t m; (void)a; (void)m; return (int)TUTEL_AMD_ABI_VERSION - 1; }\n')
    r = subprocess.run(["gcc", "-std=c99", "-Wall", "-Wextra", "-pedantic", "-Werror", "-I", os.path.join(ROOT, "include"), "-c", str(src), "-o", str(tmp_path / "h.o")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
