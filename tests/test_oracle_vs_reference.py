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
