#!/bin/bash
# Collects the HBM-side PMC counters of the dominant kernel (k_mog_fused) as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each with only
# --kernel-trace beside --pmc; units are KiB.  The guide's gfx950 caveat -- FETCH_SIZE under-counts
# wide coalesced reads by exactly 2x, other widths uncalibrated -- is handled by CALIBRATING on a
# known byte count in this kernel's own access pattern: the --dense-model run keeps all five
# mixture modes live on every pixel, so K1 must read exactly 104 B/px (3 B BGR + 1 B counter +
# 100 B of planes); fetch_factor = 104*H*W / (FETCH_SIZE*1024).
# Run on the GPU box from the repo root:   bash profiles/collect_pmc.sh   -> gpurun_out/pmc_traffic.json
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
WL="${WORKLOADS:-1080p1 1080p16 4k1}"
run_pass() {   # name counter extra-args...
  local name=$1 c=$2; shift 2
  rm -rf /tmp/pmc_${name}_$c
  rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${name}_$c -o r -- python $R/bench.py "$@" --steps 60 \
      --warmup 20 --no-cpu-baseline --no-parity > /tmp/pmc_${name}_$c.json 2> /tmp/pmc_${name}_$c.err
  python $R/profiles/summarize_pmc.py /tmp/pmc_${name}_$c/r_results.db oatgpu 21 > $R/gpurun_out/pmc_${name}_$c.md   # skip frame 1 + warm-up
}
for c in FETCH_SIZE WRITE_SIZE; do
  run_pass dense4k $c --workload 4k1 --dense-model --pool 12
  for w in $WL; do run_pass $w $c --workload $w; done
done
python - <<PY
import json
R = "$R"
def val(name, c):
    for line in open(f"{R}/gpurun_out/pmc_{name}_{c}.md"):
        if "k_mog_fused" in line:
            return float(line.split("|")[4])
    raise SystemExit(f"no k_mog_fused row for {name} {c}")
dense_fetch = val("dense4k", "FETCH_SIZE")
expected = 104.0 * 3840 * 2160
factor = expected / (dense_fetch * 1024)
out = {"_calibration": dict(workload="4k1 --dense-model", FETCH_SIZE_KiB=dense_fetch,
                            WRITE_SIZE_KiB=val("dense4k", "WRITE_SIZE"), expected_read_bytes=expected,
                            fetch_factor=factor,
                            note="all 5 modes live on every pixel => K1 reads exactly 104 B/px; "
                                 "fetch_factor = expected / (FETCH_SIZE*1024) corrects the gfx950 FETCH_SIZE under-count "
                                 "for this kernel's access widths")}
for w in "$WL".split():
    f, wr = val(w, "FETCH_SIZE"), val(w, "WRITE_SIZE")
    out[w] = dict(FETCH_SIZE_KiB=f, WRITE_SIZE_KiB=wr, k_mog_fused_bytes_per_launch=factor * f * 1024 + wr * 1024,
                  note="fetch_factor*FETCH_SIZE*1024 + WRITE_SIZE*1024, averaged over all dispatches of the run")
json.dump(out, open(f"{R}/gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
