#!/bin/bash
# Evidence of round 6's shipped tree (r08*), collected on the GPU box from the repo root:   bash profiles/collect_r08.sh [tag]
# Every command runs under `timeout -k` with stdin closed.
#   1. the GPU tests
#   2. the default bench line (+ its bench_detail.json) and the line with the driver's arguments (--steps 20 --warmup 5)
#   3. rocprofv3 --kernel-trace --stats of the default command (profiles/trace_default.sh)
#   4. rocprofv3 --kernel-trace --stats of the dense and of the benched 4K run (back-half kernels beside K1) + the timeline
#   5. SQ-side counters of the benched k_mog_fused
#   6. `python bench.py --gpus 8 --steps 20 --warmup 5 --backend gloo` as a PLAIN process on the one GPU: the driver's own multi-GPU
#      command (the whole N > 1 record: both sizes, frac_benched, per-rank kernel times, both scatter legs)
TAG=${1:-r08z}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
( timeout -k 5 1800 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-400 ) < /dev/null > $O/${TAG}_gputests.txt 2>&1
cat $O/${TAG}_gputests.txt
timeout -k 5 900 python bench.py --detail-out $O/${TAG}_bench_default_detail.json < /dev/null > $O/${TAG}_bench_default_line.json 2> $O/${TAG}_bench_default.log
wc -c $O/${TAG}_bench_default_line.json; cat $O/${TAG}_bench_default_line.json
timeout -k 5 900 python bench.py --steps 20 --warmup 5 --detail-out $O/${TAG}_bench_driver_args_detail.json < /dev/null > $O/${TAG}_bench_driver_args_line.json 2> $O/${TAG}_bench_driver_args.log
cat $O/${TAG}_bench_driver_args_line.json
timeout -k 5 900 bash profiles/trace_default.sh $TAG > /dev/null 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp
for leg in dense sparse; do
  args="--workload 4k1 --steps 300 --warmup 100 --quick --no-parity --no-spin-up"
  [ $leg = dense ] && args="$args --dense-model" || args="$args --no-dense-leg"
  rm -rf /tmp/kt_$leg
  timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$leg -o r -- python $R/bench.py $args --detail-out $O/${TAG}_bench_4k1_${leg}_traced.json > /dev/null 2> /tmp/kt_$leg.err < /dev/null
  db=$(find /tmp/kt_$leg -name "*.db" | head -1)
  {
    echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py $args"
    echo
    echo "Per-kernel statistics of this library's kernels, first 101 dispatches of every kernel skipped (frame 1 + warm-up):"
    echo
    python $R/profiles/summarize_rocpd.py $db 101
    echo
    echo "Timeline of the pipelined steps (tools/timeline.py):"
    echo
    python $R/tools/timeline.py $db 200
    echo
    echo "bench line of the traced run (HIP-event time of the same kernel on its own stream):"
    python - "$O/${TAG}_bench_4k1_${leg}_traced.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
b = j["roofline"]["benched_workload"]
print(f"  value {j['value']:.1f} fps, ms_per_step {j['ms_per_step']:.4f}, k_mog_fused avg_launch_ms (events) {b['avg_launch_ms']:.4f}, stage_ms {j['stage_ms']}, latency_us {j.get('latency_us')}")
PY
  } > $O/${TAG}_kernel_stats_4k1_${leg}.md
done
cd $R
timeout -k 5 400 bash tools/pmc_sq.sh gpurun_out/${TAG}_sq_sparse --workload 4k1 --steps 40 --warmup 100 --k1-wg 64 > /dev/null 2>&1 < /dev/null
# the driver's own multi-GPU command (+ the backend: eight ranks share the one GPU): the whole N > 1 record in one line
env -u WORLD_SIZE -u RANK -u LOCAL_RANK timeout -k 5 1000 python bench.py --gpus 8 --steps 20 --warmup 5 --backend gloo --detail-out $O/${TAG}_bench_gpus8_gloo_detail.json < /dev/null > $O/${TAG}_bench_gpus8_gloo_line.json 2> $O/${TAG}_bench_gpus8_gloo.log
cat $O/${TAG}_bench_gpus8_gloo_line.json
ls -la $O | grep ${TAG}_
