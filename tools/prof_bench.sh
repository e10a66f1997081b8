#!/bin/bash
# rocprofv3 passes over the default bench command (run on the GPU box):
#   1. --kernel-trace --stats   -> per-kernel durations (profiles/<tag>_bench_kernel_stats.csv)
#   2. --pmc FETCH_SIZE, 3. --pmc WRITE_SIZE (own passes: TCC slots) -> HBM bytes per dispatch
# usage: bash tools/prof_bench.sh <tag> [bench args...]
tag=${1:-r01}; shift
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/bench_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python bench.py --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline "$@" > $out/bench_fetch.json 2> $out/bench_fetch.err
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python bench.py --steps 16 --warmup 2 --no-cpu-baseline "$@" > $out/bench_write.json 2> $out/bench_write.err
cp $out/trace/t_kernel_stats.csv $out/kernel_stats.csv
python tools/pmc_summary.py $out > $out/summary.txt 2>&1
grep -E "fused|tiled|begin" $out/kernel_stats.csv | cut -c1-75,120-260
grep -E "fused|tiled" $out/summary.txt
python tools/trace_levels.py $out/trace/t_kernel_trace.csv 80 | head -4
tail -c 1500 $out/bench.json
find $out -name '*.csv' -size +6M -delete
