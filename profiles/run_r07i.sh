#!/bin/bash
# r07i: full validation of the shipped tree (GPU tests, default bench line, the driver's arguments) + the dense one-vs-two-frame counter table
R=${GRAFT_REPO_ROOT:-$PWD}; O=$R/gpurun_out; mkdir -p $O; cd $R
( timeout -k 5 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | cut -c1-400 ) < /dev/null > $O/r07i_gputests.txt 2>&1
cat $O/r07i_gputests.txt
timeout -k 5 900 python bench.py < /dev/null > $O/r07i_bench_default.line 2> $O/r07i_bench_default.log
cp bench_detail.json $O/r07i_bench_default_detail.json; wc -c $O/r07i_bench_default.line; cat $O/r07i_bench_default.line
timeout -k 5 900 python bench.py --steps 20 --warmup 5 < /dev/null > $O/r07i_bench_driver_args.line 2> $O/r07i_bench_driver_args.log
cp bench_detail.json $O/r07i_bench_driver_args_detail.json; cat $O/r07i_bench_driver_args.line
timeout -k 5 1200 bash tools/dense_one_vs_two.sh gpurun_out/r07i_dense_one_vs_two_frames.md < /dev/null > /dev/null 2>&1
cat $O/r07i_dense_one_vs_two_frames.md | cut -c1-200
