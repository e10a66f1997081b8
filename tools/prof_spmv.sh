#!/bin/bash
# rocprofv3 passes over the full-pass SpMV workload (run on the GPU box):
#   1. --kernel-trace --stats        -> per-kernel durations
#   2. --pmc FETCH_SIZE              -> HBM read KiB per dispatch   (own pass: TCC slots)
#   3. --pmc WRITE_SIZE              -> HBM write KiB per dispatch
# usage: bash tools/prof_spmv.sh <tag> [scale]
tag=${1:-r01}; scale=${2:-22}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/spmv_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python tools/run_spmv.py $scale 20 > $out/run.log 2>&1
rocprofv3 --pmc FETCH_SIZE --output-format csv -d $out/fetch -o f -- python tools/run_spmv.py $scale 5 > $out/run_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE --output-format csv -d $out/write -o w -- python tools/run_spmv.py $scale 5 > $out/run_write.log 2>&1
find $out -name '*kernel_stats.csv' | head -1 | xargs -I{} cp {} $out/kernel_stats.csv
python tools/pmc_summary.py $out > $out/summary.txt 2>&1
cat $out/run.log | tail -3; head -8 $out/kernel_stats.csv; cat $out/summary.txt
find $out -name '*.csv' -size +4M -delete
