#!/bin/bash
# r07in: a lone frame's back half queued on stream A itself (OATGPU_LONE_PLAIN=1, now also inline) against the plain order on a B stream (=0):
# one-frame-at-a-time latency through the library and through the process pipeline; then the GPU tests
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
{
for rep in 1 2; do for w in 4k1 1080p1 vga1 1080p2; do for lp in 0 1; do
  OATGPU_LONE_PLAIN=$lp OATGPU_MEASURE_PY=1 OATGPU_LIB=$R/build/variants/liboatgpu_meas.so timeout -k 5 300 python bench.py --workload $w --steps 600 --quick --check-steps 8 --detail-out $O/r07in_tmp.json > /dev/null 2> $O/r07in_tmp.log < /dev/null
  python - $O/r07in_tmp.json $w $lp <<'PY'
import json, sys
try:
    j = json.load(open(sys.argv[1])); l = j.get("latency_us") or {}
    print(f"{sys.argv[2]:7s} lone frame {'inline on A' if sys.argv[3] == '1' else 'early / B stream'}: fps {j['value']:9.1f}  one frame at a time p50 {l.get('single_p50'):.1f} p99 {l.get('single_p99'):.1f} us  parity {j['parity']}")
except Exception as e:
    print(sys.argv[2:], "FAILED", e)
PY
done; done; done
} > $O/r07in_lone_frame_inline_ab.txt 2>&1
cat $O/r07in_lone_frame_inline_ab.txt
( timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | cut -c1-300 ) < /dev/null > $O/r07in_gputests.txt 2>&1
cat $O/r07in_gputests.txt
timeout -k 5 600 python bench.py --workload 1080p1 --steps 300 --no-pmc --no-extra --no-cpu-baseline --no-dense-leg --detail-out $O/r07in_pipe.json > /dev/null 2>&1
python -c "
import json; j=json.load(open('$O/r07in_pipe.json')); print('pipeline:', json.dumps(j['pipeline']['track_1080p_latency'])[:600])" | tee -a $O/r07in_lone_frame_inline_ab.txt
