"""Dev tool: rocprofv3 PMC passes (one counter group per run, no trace domains -- gpurun refuses --pmc mixed with
trace domains) of one command, then a per-kernel table of per-dispatch averages.

    python tools/pmc_run.py <outdir> <kernel-name-substring> -- <command ...>

Counter groups are chosen to fit the per-block slot limits (SQ 8, TCC 4, GRBM 2; MI355X_MICROARCH.md)."""
import csv
import glob
import os
import subprocess
import sys

GROUPS = [
    ["GRBM_GUI_ACTIVE", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"],
    ["SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_VALU_MFMA_MOPS_BF16", "SQ_WAIT_INST_LDS", "SQ_ACTIVE_INST_LDS", "SQ_INSTS_LDS", "SQ_ACTIVE_INST_VALU"],
    ["SQ_LDS_BANK_CONFLICT", "SQ_LDS_IDX_ACTIVE", "SQ_LDS_ADDR_CONFLICT", "SQ_LDS_UNALIGNED_STALL", "SQ_INST_CYCLES_VMEM", "SQ_ACTIVE_INST_VMEM"],
]


def main():
    out, pat = sys.argv[1], sys.argv[2]
    cmd = sys.argv[sys.argv.index("--") + 1:]
    os.makedirs(out, exist_ok=True)
    env = dict(os.environ, TMPDIR="/tmp")
    table = {}
    for gi, grp in enumerate(GROUPS):
        d = os.path.join(out, f"g{gi}")
        r = subprocess.run(["rocprofv3", "--pmc"] + grp + ["--output-format", "csv", "-d", d, "--"] + cmd,
                           env=env, capture_output=True, text=True)
        if r.returncode != 0:
            print("pass", gi, "failed:", r.stderr[-500:])
            continue
        for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
            for row in csv.DictReader(open(f)):
                if pat in row["Kernel_Name"]:
                    k = (row["Kernel_Name"][:90], row["Counter_Name"])
                    v = table.setdefault(k, [0.0, 0])
                    v[0] += float(row["Counter_Value"])
                    v[1] += 1
    with open(os.path.join(out, "summary.txt"), "w") as fo:
        for (kn, cn), (s, n) in sorted(table.items()):
            line = f"{kn:90s} {cn:32s} {s / n:16.0f}  (n={n})"
            print(line)
            fo.write(line + "\n")


if __name__ == "__main__":
    main()
