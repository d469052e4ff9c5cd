#!/bin/bash
# Collects the HBM-side PMC counters of the dominant kernel exactly as MI355X_MICROARCH.md
# prescribes: FETCH_SIZE and WRITE_SIZE in SEPARATE passes (TCC slots), each with only
# --kernel-trace beside --pmc; units are KiB; on gfx950 FETCH_SIZE reports half the bytes of a
# wide (16 B/lane) coalesced read stream, so it is doubled.  Run on the GPU box from the repo root:
#     bash profiles/collect_pmc.sh            -> gpurun_out/pmc_traffic.json (+ per-pass tables)
set -e
R=${GRAFT_REPO_ROOT:-$PWD}
mkdir -p $R/gpurun_out
cd /tmp && export TMPDIR=/tmp
for w in ${WORKLOADS:-1080p1 1080p16 4k1}; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/pmc_${w}_$c
    rocprofv3 --kernel-trace --pmc $c -d /tmp/pmc_${w}_$c -o r -- python $R/bench.py --workload $w --steps 60 \
        --warmup 20 --no-cpu-baseline --no-parity > /tmp/pmc_${w}_$c.json 2> /tmp/pmc_${w}_$c.err
    python $R/profiles/summarize_pmc.py /tmp/pmc_${w}_$c/r_results.db > $R/gpurun_out/pmc_${w}_$c.md
  done
done
python - <<PY
import json, re
out = {}
for w in "${WORKLOADS:-1080p1 1080p16 4k1}".split():
    v = {}
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        for line in open(f"$R/gpurun_out/pmc_{w}_{c}.md"):
            if "k_mog_fused" in line:
                v[c] = float(line.split("|")[4])
    fetch_b = 2.0 * v["FETCH_SIZE"] * 1024          # gfx950 correction for 16 B/lane streams
    write_b = v["WRITE_SIZE"] * 1024
    out[w] = dict(FETCH_SIZE_KiB=v["FETCH_SIZE"], WRITE_SIZE_KiB=v["WRITE_SIZE"],
                  k_mog_fused_bytes_per_launch=fetch_b + write_b,
                  note="2*FETCH_SIZE*1024 + WRITE_SIZE*1024, averaged over all dispatches of the run")
json.dump(out, open("$R/gpurun_out/pmc_traffic.json", "w"), indent=1)
print(json.dumps(out, indent=1))
PY
