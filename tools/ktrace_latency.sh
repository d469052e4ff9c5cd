#!/bin/bash
# kernel trace of synchronous single-frame steps: tools/ktrace_latency.sh OUTFILE [latency_probe args]
out=$1; shift
R=$PWD; mkdir -p $(dirname $R/$out)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_lat
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_lat -o r -- python $R/tools/latency_probe.py "$@" > /tmp/kt_lat.out 2> /tmp/kt_lat.err || tail -3 /tmp/kt_lat.err
db=$(find /tmp/kt_lat -name "*.db" | head -1)
{ echo "# synchronous single-frame steps: $@"; cat /tmp/kt_lat.out; echo; python $R/profiles/summarize_rocpd.py $db 60 oatgpu; } > $R/$out
cat $R/$out
