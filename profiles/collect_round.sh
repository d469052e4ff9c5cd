#!/bin/bash
# Collects everything the round's committed measurements come from, on the GPU box:
#   bash profiles/collect_round.sh r01_d        -> gpurun_out/r01_d_*  (+ gpurun_out/pmc_traffic.json)
# 1. bench JSON lines (un-traced) for the three device-input workloads, the dense-model 4K case and
#    the PCIe-inclusive host-input runs;
# 2. rocprofv3 --kernel-trace --stats summaries of the same command (tracing on), per workload;
# 3. the PMC passes for HBM traffic (profiles/collect_pmc.sh: --pmc only beside --kernel-trace);
# 4. the native (no Python) loop over the same C ABI, for the host-overhead comparison.
# Afterwards copy gpurun_out/<tag>_* and pmc_traffic.json into profiles/.
TAG=${1:-r01_x}
R=${GRAFT_REPO_ROOT:-$PWD}
O=$R/gpurun_out
mkdir -p $O
cd /tmp && export TMPDIR=/tmp

bash $R/profiles/collect_pmc.sh > $O/${TAG}_pmc.log 2>&1
cp $O/pmc_traffic.json $R/profiles/pmc_traffic.json        # so that the bench lines below quote this run's traffic
{
  echo "# $TAG: HBM traffic of k_mog_fused from PMC counters (profiles/collect_pmc.sh)"
  echo
  echo '```'; cat $O/pmc_traffic.json; echo '```'
  for f in $O/pmc_*_FETCH_SIZE.md $O/pmc_*_WRITE_SIZE.md; do echo; echo "## $(basename $f .md)"; echo; cat $f; done
} > $O/${TAG}_pmc_hbm.md

for w in 1080p1 1080p16 4k1; do
  python $R/bench.py --workload $w --steps 2000 --warmup 200 2> $O/${TAG}_bench_$w.err | tail -1 > $O/${TAG}_bench_$w.json
done
python $R/bench.py --workload 4k1 --dense-model --pool 12 --steps 500 2>/dev/null | tail -1 > $O/${TAG}_bench_4k1_dense_model.json
python $R/bench.py --workload 1080p1 --learning-rate 0 2>/dev/null | tail -1 > $O/${TAG}_bench_1080p1_alpha0.json     # Oat's default -a 0: frozen model
python $R/bench.py --workload 4k1 --learning-rate 0 2>/dev/null | tail -1 > $O/${TAG}_bench_4k1_alpha0.json
for w in 1080p1 4k1; do
  python $R/bench.py --workload $w --input host --steps 600 --warmup 100 2>/dev/null | tail -1 > $O/${TAG}_bench_${w}_host_input.json
  python $R/bench.py --workload $w --input host-sync --steps 400 --warmup 50 2>/dev/null | tail -1 > $O/${TAG}_bench_${w}_host_sync_input.json
done

{
  echo "# $TAG: rocprofv3 --kernel-trace --stats"
  echo
  echo 'Command (MI355X box, from /tmp): `rocprofv3 --kernel-trace --stats -d /tmp/kt_<w> -o r -- python bench.py --workload <w> --no-cpu-baseline --no-parity` (300 timed steps, 50 warm-up);'
  echo 'table = `profiles/summarize_rocpd.py r_results.db 51` (frame 1 + warm-up dispatches skipped, only this library'"'"'s kernels).'
  echo 'The JSON line printed by the SAME (traced) run is shown first: compare its `roofline.avg_launch_ms` with the k_mog_fused row.'
  for w in 1080p1 1080p16 4k1; do
    rm -rf /tmp/kt_$w
    rocprofv3 --kernel-trace --stats -d /tmp/kt_$w -o r -- python $R/bench.py --workload $w --steps 300 --warmup 50 \
        --no-cpu-baseline --no-parity > /tmp/kt_$w.json 2> /tmp/kt_$w.err
    echo; echo "## workload $w"; echo; echo '```'; tail -1 /tmp/kt_$w.json; echo '```'; echo
    python $R/profiles/summarize_rocpd.py /tmp/kt_$w/r_results.db 51
  done
} > $O/${TAG}_kernel_stats.md 2>&1

{
  for a in "1080 1920 1 4000 3 7" "1080 1920 16 600 3 7" "2160 3840 1 3000 7 7"; do $R/build/bin/bench_native $a; done
} > $O/${TAG}_native_loop.jsonl 2>&1
ls -la $O | tail -30
