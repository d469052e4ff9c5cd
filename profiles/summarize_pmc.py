#!/usr/bin/env python3
"""Per-kernel average of each collected PMC counter from a rocpd SQLite file
(`rocprofv3 --kernel-trace --pmc <COUNTER> ...`).
Usage: summarize_pmc.py results.db [name_regex] [skip_first_n_dispatches_per_kernel]"""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    only = sys.argv[2] if len(sys.argv) > 2 else "oatgpu"
    skip = int(sys.argv[3]) if len(sys.argv) > 3 else 0
    rows = db.execute("select kernel_name, counter_name, value, duration, start from counters_collection "
                      "order by start").fetchall()
    per = {}
    for k, c, v, d, _ in rows:
        if not re.search(only, k):
            continue
        per.setdefault((k.split("(")[0], c), []).append((v, d))
    print("| kernel | counter | dispatches | avg value | avg duration us |")
    print("|---|---|---|---|---|")
    for (k, c), lst in sorted(per.items()):
        lst = lst[skip:]
        if not lst:
            continue
        n = len(lst)
        print(f"| `{k}` | {c} | {n} | {sum(v for v, _ in lst) / n:.1f} | {sum(d for _, d in lst) / n / 1e3:.2f} |")


if __name__ == "__main__":
    main()
