#!/bin/bash
# rocprofv3 --kernel-trace --stats of THE DEFAULT COMMAND (python bench.py; only the nested rocprofv3 PMC child passes
# are left out: --no-pmc), summarised per instantiation of the per-pixel kernel and launch geometry, beside the line
# the traced run printed:   bash profiles/trace_default.sh [tag]   -> gpurun_out/<tag>_trace_default.md
TAG=${1:-r03}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_default
timeout 900 rocprofv3 --kernel-trace --stats -d /tmp/kt_default -o r -- python $R/bench.py --no-pmc --detail-out $O/${TAG}_bench_default_traced.json > $O/${TAG}_bench_default_traced.line 2> /tmp/kt_default.err
db=$(find /tmp/kt_default -name "*.db" | head -1)
python - "$db" "$O/${TAG}_bench_default_traced.json" > $O/${TAG}_trace_default.md <<'PY'
import json, sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute(
    "select s.kernel_name, d.start, d.end, d.grid_size_x, d.grid_size_y "
    "from rocpd_kernel_dispatch d join rocpd_info_kernel_symbol s on d.kernel_id = s.id order by d.start").fetchall()
per = {}
for name, st, en, gx, gy in rows:
    if "k_mog_fused" not in name:
        continue
    per.setdefault((name.split("(")[0], gx, gy), []).append((en - st) / 1e3)
print("# rocprofv3 --kernel-trace --stats -- python bench.py --no-pmc   (the default command; per instantiation and grid of k_mog_fused)")
print()
print("One instantiation at one grid serves several legs of the run (ageing, spin-up on a scratch context, warm-up, the timed")
print("steps; the dense leg's burst and sustained windows): the MEDIAN is the figure to hold against the line's HIP-event")
print("averages of the timed steps; tracing itself slows the host-bound 1080p legs.")
print()
print("| kernel | grid (x, y) threads | dispatches | avg us | median us | p10 | p90 | min | max |")
print("|---|---|---|---|---|---|---|---|---|")
for (name, gx, gy), d in sorted(per.items(), key=lambda kv: -sum(kv[1])):
    q = sorted(d)
    print(f"| `{name}` | ({gx}, {gy}) | {len(d)} | {sum(d)/len(d):.1f} | {q[len(q)//2]:.1f} | {q[len(q)//10]:.1f} | {q[(9*len(q))//10]:.1f} | {min(d):.1f} | {max(d):.1f} |")
j = json.load(open(sys.argv[2]))          # the run's bench_detail (--detail-out): everything it measured
r = j["roofline"]
print()
print("The line this traced run printed (HIP events on the kernel's own stream):")
print()
print(f"* value {j['value']:.0f} fps, ms_per_step {j['ms_per_step']:.4f}, frames_per_launch {j['config']['frames_per_launch']}")
print(f"* roofline.avg_launch_ms {r['avg_launch_ms']*1e3:.1f} us (dense leg, `k_mog_fused<3, false, true, 2>` at grid (8294400, 1)), frac {r['frac']:.3f}, "
      f"burst {r.get('avg_launch_ms_burst', 0)*1e3:.1f} us")
print(f"* roofline.one_frame_a_launch.avg_launch_ms {r['one_frame_a_launch']['avg_launch_ms']*1e3:.1f} us (`k_mog_fused<3, false, true, 1>`)")
b = r["benched_workload"]
print(f"* benched_workload.avg_launch_ms {b['avg_launch_ms']*1e3:.1f} us (`k_mog_fused<3, false, false, 2>` at grid (8294400, 1))")
o = j.get("one_frame_a_launch") or {}
print(f"* one_frame_a_launch.k_mog_fused_ms {o.get('k_mog_fused_ms', 0)*1e3:.1f} us (`k_mog_fused<3, false, false, 1>` at grid (8294400, 1)), {o.get('value', 0):.0f} fps")
for k, v in (j.get("extra_workloads") or {}).items():
    print(f"* {k}: k_mog_fused_ms {v['k_mog_fused_ms']*1e3:.1f} us, {v['value']:.0f} fps")
PY
cat $O/${TAG}_trace_default.md
