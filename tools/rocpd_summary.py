"""Summarise a rocprofv3 rocpd sqlite DB: per-kernel stats (+ optional timeline of one window).
usage: python tools/rocpd_summary.py <results.db> [--timeline N]"""
import sqlite3
import sys


def main():
    db = sys.argv[1]
    c = sqlite3.connect(db)
    rows = list(c.execute("select name, start, end from kernels order by start"))
    stats = {}
    for n, s, e in rows:
        d = stats.setdefault(n, [0, 0.0, 1e18, 0.0])
        d[0] += 1
        d[1] += (e - s)
        d[2] = min(d[2], e - s)
        d[3] = max(d[3], e - s)
    total = sum(v[1] for v in stats.values())
    print("%-72s %6s %10s %10s %10s %7s" % ("kernel", "calls", "avg_us", "min_us", "max_us", "%time"))
    for n, v in sorted(stats.items(), key=lambda kv: -kv[1][1]):
        print("%-72s %6d %10.2f %10.2f %10.2f %6.1f%%" % (n[:72], v[0], v[1] / v[0] / 1e3, v[2] / 1e3, v[3] / 1e3, 100 * v[1] / total))
    if "--timeline" in sys.argv:
        k = int(sys.argv[sys.argv.index("--timeline") + 1])
        tail = rows[-k:]
        t0 = tail[0][1]
        prev_end = t0
        print("\nlast %d dispatches: start_us  dur_us  gap_before_us  kernel" % k)
        for n, s, e in tail:
            print("%10.1f %8.1f %8.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, (s - prev_end) / 1e3, n[:70]))
            prev_end = e


if __name__ == "__main__":
    main()
