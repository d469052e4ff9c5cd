#!/usr/bin/env python3
"""The steady-state timeline of a pipelined run out of a rocprofv3 --kernel-trace database: for every step (= one launch of
the per-pixel kernel) the kernel's duration, the GAP on its stream before the next launch starts, and -- for the step's
row scans and blob workgroups -- when they start and end relative to the END of their per-pixel launch.

    python tools/timeline.py results.db [skip first n steps]

Matches row scans / blob launches to the latest per-pixel launch that ENDED before they ended (early order: the blob
workgroup starts long before its row scan -- its start says when it was parked, its end when the result was written)."""
import bisect
import re
import sqlite3
import statistics as st
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    rows = db.execute("select s.kernel_name, d.start, d.end from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s "
                      "on d.kernel_id = s.id order by d.start").fetchall()
    k1 = [(s, e) for n, s, e in rows if "k_mog_fused" in n][skip:]
    if len(k1) < 10:
        print("too few per-pixel launches")
        return
    t_lo, t_hi = k1[0][0], k1[-1][1]
    rs = [(s, e) for n, s, e in rows if "k_rowscan" in n and s >= t_lo and e <= t_hi]
    bl = [(s, e) for n, s, e in rows if "k_blob_lds" in n and e >= t_lo and e <= t_hi]
    ends = [e for _, e in k1]
    us = lambda x: x / 1e3
    med = lambda v: st.median(v) if v else float("nan")
    p90 = lambda v: sorted(v)[int(0.9 * len(v))] if v else float("nan")
    dur = [us(e - s) for s, e in k1]
    period = [us(k1[i + 1][0] - k1[i][0]) for i in range(len(k1) - 1)]
    gap = [us(k1[i + 1][0] - k1[i][1]) for i in range(len(k1) - 1)]
    print(f"per-pixel kernel: {len(k1)} launches; duration median {med(dur):.1f} us (p90 {p90(dur):.1f}); period {med(period):.1f} us "
          f"(p90 {p90(period):.1f}); gap before the next launch {med(gap):.1f} us (p90 {p90(gap):.1f})")

    def rel(lst, which):
        out_s, out_e, out_d = [], [], []
        for s, e in lst:
            i = bisect.bisect_right(ends, s if which == "rs" else e) - 1     # the launch that had ended when the row scan started / result came
            if which == "bl":
                # the blob launch of step i ends behind its row scans, which start behind K1(i)'s end: latest K1 end before ITS end, minus a step if closer than a row scan
                i = bisect.bisect_right(ends, e) - 1
            if i < 0:
                continue
            out_s.append(us(s - ends[i])); out_e.append(us(e - ends[i])); out_d.append(us(e - s))
        return out_s, out_e, out_d
    s_, e_, d_ = rel(rs, "rs")
    print(f"row scan: {len(rs)} launches; starts {med(s_):.1f} us (p90 {p90(s_):.1f}) behind the end of the latest finished per-pixel launch, "
          f"runs {med(d_):.1f} us (p90 {p90(d_):.1f}), ends {med(e_):.1f} us (p90 {p90(e_):.1f}) behind it")
    s_, e_, d_ = rel(bl, "bl")
    print(f"blob workgroups: {len(bl)} launches; trace duration {med(d_):.1f} us (p90 {p90(d_):.1f}; early order: includes the time parked); "
          f"end {med(e_):.1f} us (p90 {p90(e_):.1f}) behind the end of the latest per-pixel launch that finished before them")


if __name__ == "__main__":
    main()
