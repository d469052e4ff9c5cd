#!/usr/bin/env python3
"""SYNCHRONOUS steps (one frame in, wait, one result out) out of a kernel trace: per step the span from the per-pixel kernel's start to
the blob kernel's end, each kernel's duration and the gaps between them -- against the host's time per step (tools/latency_probe.py).

    python tools/sync_step_timeline.py results.db [skip first n steps]"""
import sqlite3
import statistics as st
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    steps, cur = [], None
    for n, s, e in rows:
        if "k_mog_fused" in n:
            cur = dict(k1=(s, e), rs=None, bl=None)
            steps.append(cur)
        elif cur is not None and "k_rowscan" in n and cur["rs"] is None:
            cur["rs"] = (s, e)
        elif cur is not None and "k_blob_lds" in n and cur["bl"] is None:
            cur["bl"] = (s, e)
    steps = [x for x in steps[skip:] if x["rs"] and x["bl"]]
    us = lambda x: x / 1e3
    m = lambda v: st.median(v)
    print(f"{len(steps)} synchronous steps; median us: per-pixel kernel {m([us(x['k1'][1] - x['k1'][0]) for x in steps]):.1f}, "
          f"gap to the row scan {m([us(x['rs'][0] - x['k1'][1]) for x in steps]):.1f}, row scan {m([us(x['rs'][1] - x['rs'][0]) for x in steps]):.1f}, "
          f"gap to the blob kernel {m([us(x['bl'][0] - x['rs'][1]) for x in steps]):.1f}, blob kernel {m([us(x['bl'][1] - x['bl'][0]) for x in steps]):.1f}, "
          f"first start -> last end {m([us(x['bl'][1] - x['k1'][0]) for x in steps]):.1f}; "
          f"step to step (start of one per-pixel kernel to the next) {m([us(steps[i + 1]['k1'][0] - steps[i]['k1'][0]) for i in range(len(steps) - 1)]):.1f}")


if __name__ == "__main__":
    main()
