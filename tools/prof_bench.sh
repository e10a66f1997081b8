#!/bin/bash
# rocprofv3 kernel-trace summary of the default bench command (run on the GPU box).
# usage: bash tools/prof_bench.sh <tag> [bench args...]
tag=${1:-r01}; shift
cd /tmp && export TMPDIR=/tmp
out=$GRAFT_REPO_ROOT/gpurun_out/prof_$tag
rm -rf $out; mkdir -p $out
cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out -o bench -- python bench.py --steps 64 --warmup 8 --no-cpu-baseline "$@" > $out/bench.json 2> $out/bench.err
find $out -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv; ls -R $out | head -20
head -30 $out/kernel_stats.csv
tail -1 $out/bench.json | cut -c1-400
