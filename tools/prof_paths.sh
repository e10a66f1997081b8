#!/bin/bash
# rocprofv3 --kernel-trace --stats over the secondary measurements (tools/bench_paths.py all): per-kernel durations of
# the merge, k-hop, reach, PageRank and host-layer paths -> gpurun_out/paths_<tag>/kernel_stats.csv, and the plain
# JSON lines -> gpurun_out/paths_<tag>/bench_paths.jsonl.  Copy both into profiles/.
tag=${1:-r02}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/paths_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
rocprofv3 --kernel-trace --stats --output-format csv -d $out/trace -o t -- python tools/bench_paths.py all > $out/bench_paths_under_rocprof.jsonl 2> $out/err_rocprof.txt
cp $out/trace/t_kernel_stats.csv $out/kernel_stats.csv
python tools/bench_paths.py all > $out/bench_paths.jsonl 2> $out/err.txt
python bench.py --leg bfs --scale 24 --steps 16 --warmup 4 > $out/bench_scale24.json 2>> $out/err.txt
python tools/time_transpose.py 22 > $out/transpose22.txt 2>> $out/err.txt
cut -c1-60,100-230 $out/kernel_stats.csv | head -40
cat $out/bench_paths.jsonl | cut -c1-700
tail -c 1200 $out/bench_scale24.json
cat $out/transpose22.txt | tail -5
find $out -name '*.csv' -size +6M -delete
