#!/usr/bin/env python3
"""Prints the numbers table of DESIGN.md section 6 from the round's evidence files (profiles/r03_bench_default.json,
r03_bench_driver_args.json): python tools/design_numbers.py"""
import json
import os

R = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
d = json.load(open(os.path.join(R, "profiles", "r03_bench_default.json")))
a = json.load(open(os.path.join(R, "profiles", "r03_bench_driver_args.json")))
r, b = d["roofline"], d["roofline"]["benched_workload"]
o = d["one_frame_a_launch"]
e = d["extra_workloads"]
ab = r.get("all_background") or {}
us = lambda ms: f"{ms * 1e3:.1f}"
print("| | fps | K1 (HIP events) | notes |")
print("|---|---|---|---|")
print(f"| **`4k1`** — `value` (two frames a launch) | **{d['value']:,.0f}** (driver arguments `--steps 20 --warmup 5`: {a['value']:,.0f}; "
      f"isolated 20-step block: {a['timing']['value_isolated_block']:,.0f}) | {us(b['avg_launch_ms'])} µs per two-frame launch | "
      f"{d['ms_per_step'] * 1e3:.1f} µs per step; `frac_benched` {r['frac_benched']:.2f} (PMC {b['moved_bytes_per_px']:.1f} B/px moved, "
      f"{b['useful_bytes_per_px']:.1f} useful, waste {b['waste_ratio']:.2f}); round 2: 12 924 (driver) / 14 741–15 108 |")
print(f"| `4k1` — `value_one_frame_a_launch` | **{o['value']:,.0f}** | {us(o['k_mog_fused_ms'])} µs per frame | round 2: 9 834 |")
for k, old in (("1080p16", "62 862"), ("1080p1", "42 438")):
    print(f"| `{k}` | {e[k]['value']:,.0f} | {us(e[k]['k_mog_fused_ms'])} µs per two-frame launch | round 2: {old} |")
print(f"| dense leg (4K, all five modes live, noise ±5) | `value_dense_fps` {r['value_dense_fps']:,.0f} | **{us(r['avg_launch_ms'])} µs per TWO frames**; "
      f"one frame a launch {us(r['one_frame_a_launch']['avg_launch_ms'])} µs | `roofline.frac` **{r['frac']:.3f}** on the 208 B/px a two-frame launch must move "
      f"(PMC {r['traffic'] / 1e9:.3f} GB); `frac_one_frame` {r['frac_one_frame']:.2f}; plain copy on the same device {r['measured_stream_copy_GBps'] / 1e3:.1f} TB/s |")
if ab:
    print(f"| dense leg, every pixel stays background (noise ±3) | {ab['value_dense_fps']:,.0f} | **{us(ab['avg_launch_ms'])} µs per two frames** | "
          f"`roofline.frac_all_background` **{ab['frac']:.3f}** — the same bytes ({ab['audited_sector_bytes_per_px']:.1f} B/px audited) without `detectShadowGMM` |")
c = d["cpu_baseline"]
print()
print(f"cpu_baseline: {c['value']:.1f} fps on {c['cores']} workers, by_threads {c['by_threads']}, 1 thread {c['value_1thread']:.2f}; others "
      f"{ {k: round(v['value'], 1) for k, v in c['other_sizes'].items()} }")
p = d["pipeline"]
print(f"pipeline: mog {p['framefilt_mog_1MP']['fps']:.0f} hsv {p['posidet_hsv_1MP']['fps']:.0f} latency {p['track_1080p_latency']['free_running']['latency_us']} "
      f"{p['track_1080p_latency']['free_running']['fps']} paced {p['track_1080p_latency']['paced_500fps']['latency_us']}")
