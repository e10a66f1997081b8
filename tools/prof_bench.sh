#!/bin/bash
# rocprofv3 --kernel-trace --stats over the default bench command (run on the GPU box):
#   per-kernel durations -> gpurun_out/bench_<tag>/kernel_stats.csv (copy to profiles/<tag>_bench_kernel_stats.csv)
# The HBM-byte counters (FETCH_SIZE / WRITE_SIZE, one --pmc pass each) are collected by bench.py itself (live_pmc),
# so the plain bench run that follows carries roofline.traffic and the per-kernel table of the same build.
# usage: bash tools/prof_bench.sh <tag> [bench args...]
tag=${1:-r02}; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/bench_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --no-cpu-baseline --no-pmc "$@" > $out/bench_under_rocprof.json 2> $out/bench_under_rocprof.err
cp $out/trace/t_kernel_stats.csv $out/kernel_stats.csv
python tools/trace_levels.py $out/trace/t_kernel_trace.csv 80 | head -4
python bench.py "$@" > $out/bench.json 2> $out/bench.err
cut -c1-70,110-250 $out/kernel_stats.csv | head -30
tail -c 3000 $out/bench.json
find $out -name '*.csv' -size +6M -delete
