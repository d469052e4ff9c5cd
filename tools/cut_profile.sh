#!/bin/bash
# Dynamic instruction profile of k_mog_fused by truncation, on the GPU box:  tools/cut_profile.sh OUT [--fusion1] [--dense]
# needs the variants liboatgpu_cut1.so .. liboatgpu_cut5.so and liboatgpu_base.so (make variant NAME=cutN DEFS=-DOATGPU_CUT=N)
out=$1; shift
R=$PWD; mkdir -p $R/$out
mkdir -p /tmp/cut_state; python tools/cut_profile.py save "$@" /tmp/cut_state > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
{
echo "| cut | up to | VALU / wave | SALU / wave | VMEM rd / wave | us |"
echo "|---|---|---|---|---|---|"
for n in 1 2 3 4 5 base; do
  lib=$R/build/variants/liboatgpu_cut$n.so; [ $n = base ] && lib=$R/build/variants/liboatgpu_base.so
  rm -rf /tmp/cutp
  OATGPU_MEASURE_PY=1 OATGPU_LIB=$lib timeout 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_WAVES -d /tmp/cutp -o r -- python $R/tools/cut_profile.py run "$@" /tmp/cut_state > /dev/null 2> /tmp/cutp.err || tail -2 /tmp/cutp.err
  db=$(find /tmp/cutp -name "*.db" | head -1)
  python - $db $n <<'PY'
import sqlite3, sys
db = sqlite3.connect(sys.argv[1])
rows = db.execute("select kernel_name, counter_name, value, duration from counters_collection order by start").fetchall()
per = {}
for k, c, v, d in rows:
    if "k_mog_fused" in k:
        per.setdefault(c, []).append((v, d))
names = {"1": "phase 1 + mode 0", "2": "+ phase 2 (masks, zero-inits, loads)", "3": "+ modes 1..4", "4": "+ finish (renormalise, new mode)",
         "5": "+ HSV / inRange (and all of frame 2 with two frames a launch)", "base": "+ stores (the whole kernel)"}
def avg(c):
    l = per.get(c, [])[8:]
    return (sum(v for v, _ in l) / len(l), sum(d for _, d in l) / len(l) / 1e3) if l else (0, 0)
w = avg("SQ_WAVES")[0] or 1
print(f"| {sys.argv[2]} | {names[sys.argv[2]]} | {avg('SQ_INSTS_VALU')[0] / w:.1f} | {avg('SQ_INSTS_SALU')[0] / w:.1f} | {avg('SQ_INSTS_VMEM_RD')[0] / w:.2f} | {avg('SQ_INSTS_VALU')[1]:.1f} |")
PY
done
} | tee $R/$out/cut_profile.md
