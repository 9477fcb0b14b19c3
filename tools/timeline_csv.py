#!/usr/bin/env python3
"""Kernel timeline of one train step from a rocprofv3 --kernel-trace CSV (steps delimited by k_pack_all).
    python tools/timeline_csv.py <dir with *_kernel_trace.csv> [step_index]"""
import csv
import glob
import re
import sys

csv.field_size_limit(1 << 30)


def short(n):
    n = re.sub(r"^void ", "", n)
    return re.sub(r"\(.*$", "", n).replace("mvh::", "")[:46]


def main():
    f = glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r["Queue_Id"]))
            for r in csv.DictReader(open(f))]
    rows.sort(key=lambda r: r[1])
    idx = [i for i, r in enumerate(rows) if "k_pack_all" in r[0]]
    s = int(sys.argv[2]) if len(sys.argv) > 2 else len(idx) // 2 + 10
    a, b = idx[s], idx[s + 1]
    t0 = rows[a][1]
    for r in rows[a:b]:
        print(f"{(r[1] - t0) / 1e3:8.1f} {(r[2] - r[1]) / 1e3:7.1f} {(r[2] - t0) / 1e3:8.1f} s={r[3]} {short(r[0])}")
    print("step span", (rows[b][1] - t0) / 1e3)


if __name__ == "__main__":
    main()
