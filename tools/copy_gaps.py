#!/usr/bin/env python3
"""Timeline of the host-to-device copies of a rocprofv3 --memory-copy-trace run (rocpd database): duration of every copy and
the gap between the end of one copy and the start of the next.   python tools/copy_gaps.py results.db [min_bytes]"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    min_bytes = int(sys.argv[2]) if len(sys.argv) > 2 else 1 << 20
    tabs = [r[0] for r in db.execute("select name from sqlite_master where type in ('table','view')")]
    t = [x for x in tabs if "memory_cop" in x]
    if not t:
        print("no memory copy table:", tabs)
        return
    name = sorted(t, key=len)[0]
    cols = [r[1] for r in db.execute(f"pragma table_info({name})")]
    print("table", name, cols)
    sz = "size" if "size" in cols else ("bytes" if "bytes" in cols else None)
    rows = db.execute(f"select start, end{', ' + sz if sz else ''} from {name} order by start").fetchall()
    rows = [r for r in rows if not sz or r[2] >= min_bytes]
    if len(rows) < 10:
        print("copies:", len(rows))
        return
    rows = rows[len(rows) // 4:]                      # steady state
    dur = sorted((e - s) / 1e3 for s, e, *_ in rows)
    gaps = sorted((rows[i + 1][0] - rows[i][1]) / 1e3 for i in range(len(rows) - 1))
    per = sorted((rows[i + 1][0] - rows[i][0]) / 1e3 for i in range(len(rows) - 1))
    q = lambda a, f: a[int(f * (len(a) - 1))]
    print(f"{len(rows)} copies; duration us p10/p50/p90 {q(dur, .1):.1f}/{q(dur, .5):.1f}/{q(dur, .9):.1f}; "
          f"gap end->next start us p10/p50/p90 {q(gaps, .1):.1f}/{q(gaps, .5):.1f}/{q(gaps, .9):.1f}; "
          f"start->start us p10/p50/p90 {q(per, .1):.1f}/{q(per, .5):.1f}/{q(per, .9):.1f}")
    if sz:
        print(f"bytes per copy {rows[0][2]}; rate while copying {rows[0][2] / q(dur, .5) / 1e3:.1f} GB/s")


if __name__ == "__main__":
    main()
