#!/bin/bash
# rocprofv3 --kernel-trace of a short bench run; prints per-kernel stats.  tools/ktrace.sh OUTFILE bench-args...
out=$1; shift
R=$PWD; mkdir -p $(dirname $R/$out)
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/kt_run
timeout 300 rocprofv3 --kernel-trace -d /tmp/kt_run -o r -- python $R/bench.py --pmc-child "$@" > /dev/null 2> /tmp/kt_run.err || tail -3 /tmp/kt_run.err
db=$(find /tmp/kt_run -name "*.db" | head -1)
python $R/profiles/summarize_rocpd.py $db 60 > $R/$out
cat $R/$out
