"""Pins the oracle against the reference ITSELF, run live (python package imported from
/root/reference + its C++ CPU kernels compiled into oracle/_ref/ by oracle/Makefile).
Only possible in the build container; on the GPU box the committed fixtures do the pinning."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = "/root/reference"


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tutel")), reason="reference tree not present on this box")
def test_oracle_equals_live_reference():
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "tutel_custom_kernel.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden.py"), "--check"],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "tutel")), reason="reference tree not present on this box")
def test_oracle_expert_parallel_equals_live_reference_over_gloo():
    """The oracle's W-rank simulation (moe_forward_ep: routing per rank, all-to-all layout, experts on the received rows, return
    all-to-all, decode; cross-rank capacity for dropless and for inequivalent_tokens) against the reference RUNNING with W = 2 / 4
    ranks over gloo in this container: y, the rows each rank's experts receive, dispatch counts -- bit for bit."""
    if not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "tutel_custom_kernel.so")):
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "golden", "make_golden_ep.py"), "--check"],
                       capture_output=True, text=True, timeout=1200)
    assert r.returncode == 0 and "0 mismatches" in r.stdout, r.stdout[-3000:] + r.stderr[-3000:]


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "tutel_custom_kernel.so")) and not os.path.isdir(os.path.join(REF, "tutel")),
                    reason="neither the prebuilt oracle/_ref/ nor the reference tree is on this box")
@pytest.mark.parametrize("T,M,H,E,k,cf,post,dts", [(512, 64, 32, 16, 2, 1.0, True, "float32"), (300, 64, 32, 12, 4, 1.0, False, "float32"),
                                                    (512, 64, 32, 16, 2, 0.5, True, "float32"), (512, 64, 32, 16, 1, 0.0, True, "float32"),
                                                    (512, 64, 64, 16, 2, 1.0, True, "bfloat16"), (256, 64, 32, 8, 2, 1.0, True, "float64"),
                                                    (4096, 2048, 64, 64, 2, 1.0, True, "float32")])
def test_reference_kernels_equal_the_port(T, M, H, E, k, cf, post, dts):
    """oracle/ref_kernels.py -- fast_encode / fast_decode through the reference's OWN compiled CPU kernels (oracle/_ref/: custom_kernel.cpp
    `invoke_cpu_fp32`, called as fast_dispatch.py:16-29,52-66 call it) -- against the plain-C port in moe_oracle.c: encoded rows, layer
    output, l_aux bit for bit.  This is the pair bench.py's cpu_baseline times on the GPU box (kind "reference" with the port beside it),
    where the prebuilt .so travels and the reference's Python does not; the last case is the headline's routing shape (64 experts,
    4096 tokens, model_dim 2048) with a thin expert."""
    import torch
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    from oracle import moe_oracle as O, ref_kernels as R
    if not R.available():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    dtype = getattr(torch, dts)
    x, wg, w1, b1, w2, b2 = O.make_problem(T, M, H, E, dtype=dtype, seed=11)
    a = O.moe_forward(x, wg, w1, b1, w2, b2, top_k=k, capacity_factor=cf, is_postscore=post)
    b = R.moe_forward(x, wg, w1, b1, w2, b2, top_k=k, capacity_factor=cf, is_postscore=post)
    assert "tutel_custom_kernel" in R.module().__name__ and "oracle/_ref/tutel_custom_kernel.so" in open("/proc/self/maps").read()
    assert torch.equal(a[3]["encoded"], b[3]["encoded"]) and torch.equal(a[0], b[0]) and float(a[1]) == float(b[1])
    crit = a[2]
    enc = R.fast_encode(x.float(), crit, post)
    assert torch.equal(enc, O.fast_encode(x.float(), crit, post)) and torch.equal(R.fast_decode(enc, crit, post), O.fast_decode(enc, crit, post))


@pytest.mark.skipif(not os.path.exists(os.path.join(ROOT, "oracle", "_ref", "tutel_custom_kernel.so")) and not os.path.isdir(os.path.join(REF, "tutel")),
                    reason="neither the prebuilt oracle/_ref/ nor the reference tree is on this box")
def test_bench_cpu_baseline_times_the_reference_kernels():
    """bench.py's cpu_baseline leg (the only place outside tests/ and smoke() that may touch oracle/): with oracle/_ref/ present it
    reports kind "reference" (the reference's compiled kernels, compared bit for bit with the port before either is quoted) and carries
    the port's figure beside it; a small shape here, the headline's on the GPU box."""
    sys.path.insert(0, ROOT) if ROOT not in sys.path else None
    import bench
    from oracle import ref_kernels as R
    if not R.available():
        subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle")])
    r = bench.cpu_baseline(512, 256, 128, 16, 2, max_seconds=4.0)
    assert r["kind"] == "reference" and r["unit"] == "tokens/s" and r["value"] > 0 and r["cores"] >= 1
    assert r["port"]["kind"] == "port" and r["port"]["value"] > 0 and "oracle/_ref" in r["kind_note"] and "bit for bit" in r["sample"]
