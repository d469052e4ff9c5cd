#!/usr/bin/env python3
"""Per-kernel statistics (what `rocprofv3 --kernel-trace --stats` reports) from a rocpd SQLite
result file.  Usage: summarize_rocpd.py results.db [skip_first_n_dispatches_per_kernel] [name_regex] > summary.md"""
import sqlite3
import sys


def main():
    db = sqlite3.connect(sys.argv[1])
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
    only = sys.argv[3] if len(sys.argv) > 3 else "oatgpu|rocclr"   # regex on kernel names
    import re
    rows = db.execute(
        "select s.kernel_name, d.start, d.end, s.arch_vgpr_count, s.sgpr_count, s.group_segment_size, "
        "d.grid_size_x, d.grid_size_y, d.workgroup_size_x "
        "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
    per = {}
    for name, st, en, vg, sg, lds, gx, gy, wx in rows:
        if not re.search(only, name):
            continue
        per.setdefault(name, dict(d=[], vg=vg, sg=sg, lds=lds, grid=(gx, gy), wg=wx))["d"].append(en - st)
    tot = sum(sum(v["d"][skip:]) for v in per.values()) or 1
    print("| kernel | calls | total ms | avg us | min us | max us | % | VGPR | SGPR | LDS B | grid(x,y) threads | wg |")
    print("|---|---|---|---|---|---|---|---|---|---|---|---|")
    for name, v in sorted(per.items(), key=lambda kv: -sum(kv[1]["d"][skip:])):
        d = v["d"][skip:]
        if not d:
            continue
        short = name.split("(")[0]
        print(f"| `{short}` | {len(d)} | {sum(d)/1e6:.3f} | {sum(d)/len(d)/1e3:.2f} | {min(d)/1e3:.2f} | "
              f"{max(d)/1e3:.2f} | {100*sum(d)/tot:.1f} | {v['vg']} | {v['sg']} | {v['lds']} | {v['grid']} | {v['wg']} |")


if __name__ == "__main__":
    main()
