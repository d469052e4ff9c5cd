#!/bin/bash
# Evidence of the last session of round 4 (r05*), collected on the GPU box from the repo root:   bash profiles/collect_r05.sh [tag]
# Every command runs under `timeout -k` with stdin closed (a `cut` without a file once waited 36 GPU-minutes for its stdin).
#   1. the default bench line and the line with the driver's arguments (--steps 20 --warmup 5)
#   2. rocprofv3 --kernel-trace --stats of the default command (profiles/trace_default.sh)
#   3. rocprofv3 --kernel-trace --stats of the dense and of the benched 4K run (back-half kernels beside K1), and the
#      longest launches of the back-half kernels with what ran beside them (tools/trace_outliers.py)
#   4. SQ-side counters of k_mog_fused: two frames a launch (sparse, dense) and ONE frame a launch (sparse)
#   5. the N-camera boundary: 8 x 1080p cameras over 6000 frames, DMA and copy-kernel staging, with the tracker's own clock
# Output under gpurun_out/<tag>_*; copy what is to be judged into profiles/.
TAG=${1:-r05}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd $R
timeout -k 5 900 python bench.py < /dev/null > $O/${TAG}_bench_default.json 2> $O/${TAG}_bench_default.log
timeout -k 5 900 python bench.py --steps 20 --warmup 5 > $O/${TAG}_bench_driver_args.json 2> $O/${TAG}_bench_driver_args.log
timeout -k 5 900 bash profiles/trace_default.sh $TAG > /dev/null 2>&1 < /dev/null
cd /tmp && export TMPDIR=/tmp
for leg in dense sparse; do
  args="--workload 4k1 --steps 300 --warmup 100 --quick --no-parity --no-spin-up"
  [ $leg = dense ] && args="$args --dense-model" || args="$args --no-dense-leg"
  rm -rf /tmp/kt_$leg
  timeout -k 5 600 rocprofv3 --kernel-trace --stats -d /tmp/kt_$leg -o r -- python $R/bench.py $args > $O/${TAG}_bench_4k1_${leg}_traced.json 2> /tmp/kt_$leg.err
  db=$(find /tmp/kt_$leg -name "*.db" | head -1)
  {
    echo "# $TAG: rocprofv3 --kernel-trace --stats -- python bench.py $args"
    echo
    echo "Per-kernel statistics of this library's kernels, first 101 dispatches of every kernel skipped (frame 1 + warm-up):"
    echo
    python $R/profiles/summarize_rocpd.py $db 101
    echo
    echo "The longest launches of the back-half kernels in the WHOLE trace (nothing skipped) and what ran beside them:"
    echo
    python $R/tools/trace_outliers.py $db "k_rowscan|k_blob_lds" 3
    echo
    echo "bench line of the traced run (HIP-event time of the same kernel on its own stream):"
    python - "$O/${TAG}_bench_4k1_${leg}_traced.json" <<'PY'
import json, sys
j = json.load(open(sys.argv[1]))
b = j["roofline"]["benched_workload"]
print(f"  value {j['value']:.1f} fps, ms_per_step {j['ms_per_step']:.4f}, k_mog_fused avg_launch_ms (events) {b['avg_launch_ms']:.4f}, stage_ms {j['stage_ms']}")
PY
  } > $O/${TAG}_kernel_stats_4k1_${leg}.md
done
cd $R
timeout -k 5 600 bash tools/pmc_sq.sh gpurun_out/${TAG}_sq_sparse --workload 4k1 --steps 40 --warmup 100 > /dev/null 2>&1
timeout -k 5 600 bash tools/pmc_sq.sh gpurun_out/${TAG}_sq_dense --workload 4k1 --dense-model --steps 40 --warmup 20 > /dev/null 2>&1
timeout -k 5 600 bash tools/pmc_sq.sh gpurun_out/${TAG}_sq_sparse_one_frame --workload 4k1 --steps 40 --warmup 100 --fusion 1 > /dev/null 2>&1
for m in dma; do
  echo "== --stage-copy $m"
  timeout -k 5 300 python tools/pipeline_fps.py --rows 1080 --cols 1920 --frames 6000 --fused --cameras 8 --ring 4 --stage-copy $m --timing 2>&1 | grep -v Exiting | tail -3
done > $O/${TAG}_pipeline_8cam_timing.txt
timeout -k 5 300 python tools/contexts_probe.py > $O/${TAG}_contexts.json 2> /dev/null
ls -la $O | grep ${TAG}_
