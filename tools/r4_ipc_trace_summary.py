#!/usr/bin/env python3
"""kernel_trace.csv files of tools/r4_ipc_trace.py (one per rank process) -> ONE timeline of the last forward of every rank on the
shared GPU: start / end / duration of each kernel relative to the earliest start, which process and hardware queue it ran on."""
import csv
import glob
import sys


def short(n):
    for key, s in (("expert_gemm_pp", "GEMM pp 256x256"), ("expert_gemm_big", "GEMM big"), ("expert_gemm_glds", "GEMM 128"), ("expert_gemm_kernel", "GEMM 128 reg"),
                   ("encode_kernel", "encode (peer stores)"), ("decode_kernel", "decode"), ("ep_signal", "signal"), ("ep_wait", "wait"),
                   ("gate_topk", "top-k"), ("location_kernel", "location"), ("Cijk_", "gate GEMM (lib)")):
        if key in n:
            return s
    return n[:40]


def main():
    d = sys.argv[1]
    per = {}
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = list(csv.DictReader(open(f)))
        if not rows or not any("decode_kernel" in r["Kernel_Name"] for r in rows):
            continue
        rows.sort(key=lambda r: int(r["Start_Timestamp"]))
        per[f.split("/")[-1].split("_")[0]] = rows
    sel = []
    for pid, rows in per.items():
        names = [r["Kernel_Name"] for r in rows]
        ends = [i for i, n in enumerate(names) if "decode_kernel" in n]
        last = ends[-1]
        first = [i for i, n in enumerate(names[:last]) if n.startswith("Cijk_")][-1]
        sel += [(pid, r) for r in rows[first:last + 1]]
    t0 = min(int(r["Start_Timestamp"]) for _, r in sel)
    sel.sort(key=lambda pr: int(pr[1]["Start_Timestamp"]))
    pids = sorted(per)
    print(f"# last forward of {len(pids)} rank processes sharing one GPU; times in us from the earliest kernel start")
    print(f"# {'rank':>4s} {'start':>8s} {'end':>8s} {'dur':>7s}  queue  kernel")
    for pid, r in sel:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        print(f"  {pids.index(pid):4d} {s:8.1f} {e:8.1f} {e - s:7.1f}  {r.get('Queue_Id', '?'):>5s}  {short(r['Kernel_Name'])}")
    for i, pid in enumerate(pids):
        mine = [r for p, r in sel if p == pid]
        span = (max(int(r["End_Timestamp"]) for r in mine) - min(int(r["Start_Timestamp"]) for r in mine)) / 1e3
        busy = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in mine if "ep_wait" not in r["Kernel_Name"])
        waits = sum((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3 for r in mine if "ep_wait" in r["Kernel_Name"])
        print(f"# rank {i}: forward span {span:.1f} us, sum of kernel durations without the wait kernels {busy:.1f} us, wait kernels {waits:.1f} us")


if __name__ == "__main__":
    main()
