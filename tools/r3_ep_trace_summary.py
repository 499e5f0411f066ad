#!/usr/bin/env python3
"""kernel_trace.csv of tools/r3_ep_trace.py -> timeline of the LAST forward: start / end of every kernel relative to the forward's
first kernel, the hardware queue it ran on, and how much of the expert GEMM time ran concurrently with another expert GEMM or
with an RCCL kernel."""
import csv
import glob
import sys


def main():
    d = sys.argv[1]
    rows = []
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows += list(csv.DictReader(open(f)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    # a forward starts at the gate projection's library GEMM (Cijk...) and ends at decode_kernel
    ends = [i for i, n in enumerate(names) if n.startswith("void decode_kernel")]
    last_end = ends[-1]
    starts = [i for i, n in enumerate(names[:last_end]) if n.startswith("Cijk_")]
    first = starts[-1]
    t0 = int(rows[first]["Start_Timestamp"])
    sel = [r for r in rows if t0 <= int(r["Start_Timestamp"]) <= int(rows[last_end]["End_Timestamp"])]
    print(f"# last forward: {len(sel)} kernels, {(int(rows[last_end]['End_Timestamp']) - t0) / 1e3:.1f} us from the first start to the last end")
    print(f"# {'start_us':>9s} {'end_us':>9s} {'dur_us':>8s}  queue  kernel")
    iv = []
    for r in sel:
        s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
        n = r["Kernel_Name"]
        short = n[:90]
        print(f"  {s:9.1f} {e:9.1f} {e - s:8.1f}  {r.get('Queue_Id', '?'):>5s}  {short}")
        kind = "gemm" if "expert_gemm" in n else ("rccl" if ("ccl" in n.lower() or "AllToAll" in n or "SendRecv" in n) else "other")
        iv.append((s, e, kind))
    gem = [(s, e) for s, e, k in iv if k == "gemm"]
    tot = sum(e - s for s, e in gem)
    ov_g = sum(max(0.0, min(e1, e2) - max(s1, s2)) for i, (s1, e1) in enumerate(gem) for (s2, e2) in gem[i + 1:])
    ov_r = sum(max(0.0, min(e1, e2) - max(s1, s2)) for (s1, e1) in gem for s2, e2, k in iv if k == "rccl")
    print(f"# expert GEMM time {tot:.1f} us; of it {ov_g:.1f} us ran beside another expert GEMM, {ov_r:.1f} us beside an RCCL kernel")


if __name__ == "__main__":
    main()
