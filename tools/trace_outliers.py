#!/usr/bin/env python3
"""Which launches are a kernel's longest ones, and what ran beside them?  (the 2.7-8.1 ms maxima of
k_blob_lds / k_rowscan in a kernel trace.)

    python tools/trace_outliers.py results.db [kernel regex] [top n]

From a rocprofv3 --kernel-trace rocpd database: for the n longest dispatches of every kernel matching the regex, the index
of the dispatch among that kernel's launches, when it started (ms since the first dispatch of the run), its duration,
and every other kernel whose execution overlapped it (name, overlap in us)."""
import re
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    pat = sys.argv[2] if len(sys.argv) > 2 else "k_rowscan|k_blob_lds"
    top = int(sys.argv[3]) if len(sys.argv) > 3 else 3
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    if not rows:
        print("no dispatches")
        return
    t0 = rows[0][1]
    short = lambda n: re.sub(r"\(.*", "", n).replace("void oatgpu::", "").replace("_ZN6oatgpu", "")[:70]
    per = {}
    for i, (n, st, en) in enumerate(rows):
        per.setdefault(n, []).append((en - st, st, en, len(per.get(n, []))))
    for n, lst in per.items():
        if not re.search(pat, n):
            continue
        print(f"## {short(n)}: {len(lst)} dispatches, median {sorted(d for d, *_ in lst)[len(lst) // 2] / 1e3:.1f} us")
        for dur, st, en, idx in sorted(lst, reverse=True)[:top]:
            print(f"- dispatch #{idx} at {(st - t0) / 1e6:.2f} ms: {dur / 1e3:.1f} us; beside it:")
            over = {}
            for n2, s2, e2 in rows:
                if e2 <= st or s2 >= en or (n2 == n and s2 == st):
                    continue
                o = min(en, e2) - max(st, s2)
                k = short(n2)
                over[k] = (over.get(k, (0, 0))[0] + o, over.get(k, (0, 0))[1] + 1)
            for k, (o, c) in sorted(over.items(), key=lambda kv: -kv[1][0])[:6]:
                print(f"    {k}: {c} dispatch(es), {o / 1e3:.1f} us of overlap")


if __name__ == "__main__":
    main()
