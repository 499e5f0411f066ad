#!/usr/bin/env python3
"""Training step time of the reference's 9 golden-loss configurations (tests/test_baseline.json: helloworld, 16 x 1024 tokens,
model_dim = hidden = 2048, top-1/2, 1-2 local experts, fp16 / fp32 / fp64) on this GPU, beside the only performance numbers the
reference publishes -- the V100 / RTX 2080 Ti step times stored in that file (tests/test_baseline.json:8-13, ...; copied into
tests/golden/reference_baseline_losses.json's generator comment and BASELINE.md section 1).

Each case = `python -m tutel_amd.examples.helloworld <flags> --num_steps N` (forward, backward through the HIP dispatch /
combine kernels + gate-grad, SGD); the script's own "[Summary] Average synchronized step_time" (mean of the last 10 steps,
device-synchronised, as the reference measures it).  Prints one JSON object."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF = {  # (top, dtype, experts) -> {gpu: seconds}: the reference's own tests/test_baseline.json
    (1, "float16", 1): {"V100": 0.014220, "RTX 2080 Ti": 0.016164}, (1, "float16", 2): {"V100": 0.015885, "RTX 2080 Ti": 0.016641},
    (1, "float32", 1): {"V100": 0.068639, "RTX 2080 Ti": 0.059459}, (1, "float32", 2): {"V100": 0.070857, "RTX 2080 Ti": 0.069748},
    (2, "float16", 1): {"V100": 0.014011, "RTX 2080 Ti": 0.016270}, (2, "float16", 2): {"V100": 0.030053, "RTX 2080 Ti": 0.032551},
    (2, "float32", 1): {"V100": 0.069216, "RTX 2080 Ti": 0.059893}, (2, "float32", 2): {"V100": 0.136276, "RTX 2080 Ti": 0.140044},
    (2, "float64", 2): {"RTX 2080 Ti": 0.220799},
}


def main():
    steps = int(sys.argv[1]) if len(sys.argv) > 1 else 40
    cases = json.load(open(os.path.join(ROOT, "tests", "golden", "reference_baseline_losses.json")))["cases"]
    out = []
    for c in cases:
        r = subprocess.run([sys.executable, "-m", "tutel_amd.examples.helloworld", "--num_steps", str(steps)] + c["flags"].split(),
                           cwd=ROOT, env=dict(os.environ, MASTER_ADDR="127.0.0.1"), capture_output=True, text=True, timeout=600)
        txt = r.stdout + r.stderr
        t = None
        for ln in txt.splitlines():
            if "Average synchronized step_time" in ln:
                t = float(ln.split("=")[1].split()[0])
        key = (c["top"], c["dtype"], c["num_local_experts"])
        ref = REF.get(key, {})
        out.append({"top": c["top"], "dtype": c["dtype"], "num_local_experts": c["num_local_experts"], "mi355x_step_s": t,
                    "reference_step_s": ref, "speedup_vs_v100": round(ref["V100"] / t, 2) if t and "V100" in ref else None})
    print(json.dumps({"steps": steps, "cases": out}, indent=1))


if __name__ == "__main__":
    main()
