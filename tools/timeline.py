#!/usr/bin/env python3
"""Print the kernel timeline of one train step from a rocprofv3 --kernel-trace rocpd database.

    python tools/timeline.py gpurun_out/prof_x/x_results.db [step_index] [--stats]
Steps are delimited by k_pack_all launches."""
import re
import sqlite3
import sys


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).replace("mvh::", "")[:48]


def main():
    db = sys.argv[1]
    step = int(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2].isdigit() else None
    c = sqlite3.connect(db)
    rows = c.execute("select name,start,end,queue_id from kernels order by start").fetchall()
    idx = [i for i, r in enumerate(rows) if "k_pack_all" in r[0]]
    if "--stats" in sys.argv:
        a, b = idx[len(idx) // 4], idx[-2]
        n_steps = sum(1 for i in idx if a <= i < b)
        agg = {}
        for r in rows[a:b]:
            k = short(r[0])
            t = agg.setdefault(k, [0, 0.0])
            t[0] += 1
            t[1] += (r[2] - r[1]) / 1e3
        tot = sum(v[1] for v in agg.values())
        print(f"{'kernel':50s} calls/step  us/step   avg_us")
        for k, v in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print(f"{k:50s} {v[0] / n_steps:8.1f} {v[1] / n_steps:9.1f} {v[1] / v[0]:8.2f}")
        print(f"{'TOTAL busy':50s} {'':8s} {tot / n_steps:9.1f}   wall/step {(rows[b][1] - rows[a][1]) / 1e3 / n_steps:.1f} us")
        return
    s = step if step is not None else len(idx) // 2
    a, b = idx[s], idx[s + 1]
    t0 = rows[a][1]
    for r in rows[a:b]:
        print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} q={r[3]} {short(r[0])}")
    print("step span", (rows[b][1] - t0) / 1e3)


if __name__ == "__main__":
    main()
