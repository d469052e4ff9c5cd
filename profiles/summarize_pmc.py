#!/usr/bin/env python3
"""Per-kernel average of each collected PMC counter from a rocpd SQLite file
(`rocprofv3 --kernel-trace --pmc <COUNTER> ...`).  Usage: summarize_pmc.py results.db [name_regex]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    only = sys.argv[2] if len(sys.argv) > 2 else "oatgpu"
    rows = db.execute("select kernel_name, counter_name, value, duration from counters_collection").fetchall()
    agg = {}
    for k, c, v, d in rows:
        if not re.search(only, k):
            continue
        a = agg.setdefault((k.split("(")[0], c), [0, 0.0, 0.0])
        a[0] += 1; a[1] += v; a[2] += d
    print("| kernel | counter | dispatches | avg value | avg duration us |")
    print("|---|---|---|---|---|")
    for (k, c), (n, v, d) in sorted(agg.items()):
        print(f"| `{k}` | {c} | {n} | {v / n:.1f} | {d / n / 1e3:.2f} |")


if __name__ == "__main__":
    main()
