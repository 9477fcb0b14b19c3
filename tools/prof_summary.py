#!/usr/bin/env python3
"""Condense a rocprofv3 `--kernel-trace --stats` kernel_stats.csv into a short table
(kernel names cut at the first '(' / template list) for committing under profiles/."""
import csv
import re
import sys


def short(name):
    name = re.sub(r"^void ", "", name)
    name = re.sub(r"\(.*$", "", name)
    if len(name) > 90:
        name = name[:87] + "..."
    return name


def main(path, steps=None):
    rows = list(csv.DictReader(open(path)))
    tot = sum(int(r["TotalDurationNs"]) for r in rows)
    print("kernel,calls,total_us,avg_us,min_us,max_us,percent")
    for r in rows:
        print(f'"{short(r["Name"])}",{r["Calls"]},{int(r["TotalDurationNs"]) / 1e3:.1f},'
              f'{float(r["AverageNs"]) / 1e3:.2f},{int(r["MinNs"]) / 1e3:.2f},{int(r["MaxNs"]) / 1e3:.2f},{r["Percentage"]}')
    print(f'"TOTAL",,{tot / 1e3:.1f},,,,100')
    if steps:
        print(f'"per-step GPU busy (us) over {steps} steps",,{tot / 1e3 / int(steps):.1f},,,,')


if __name__ == "__main__":
    main(*sys.argv[1:])
