#!/bin/bash
# One rocprofv3 --kernel-trace --stats table PER LEG of the bench: the headline k-hop leg at RMAT-22,
# the same leg at 24 and 26, the BFS leg at 22 and 26.  Run on the GPU box:  bash tools/prof_legs.sh <tag>
# -> gpurun_out/legs_<tag>/<leg>_kernel_stats.csv (+ the bench line of each run); copy into profiles/.
tag=${1:-r04}
root=$GRAFT_REPO_ROOT
out=$root/gpurun_out/legs_$tag
rm -rf $out; mkdir -p $out
cd /tmp && export TMPDIR=/tmp
cd $root
run() {  # name, bench args...
  name=$1; shift
  rocprofv3 --kernel-trace --stats --output-format csv -d $out/$name -o t -- python bench.py --no-cpu-baseline --no-parity --no-pmc --no-lanes-sweep "$@" > $out/$name.out 2> $out/$name.err
  cp $out/$name/*/t_kernel_stats.csv $out/${name}_kernel_stats.csv 2>/dev/null || cp $out/$name/t_kernel_stats.csv $out/${name}_kernel_stats.csv
  tail -1 $out/$name.out | cut -c1-400
  head -6 $out/${name}_kernel_stats.csv | cut -c1-160
}
run khop22 --quick --scale 22 --steps 8 --warmup 2
# the same leg on ONE lane: kernels run alone, the averages are what bench.py's roofline (a one-lane replay) quotes
run khop22_1lane --quick --scale 22 --steps 4 --warmup 1 --opt expand_scan_lanes=1
run khop24 --quick --scale 24 --steps 8 --warmup 2 --sources-per-call 16384
run khop26 --quick --scale 26 --steps 4 --warmup 1 --sources-per-call 8192
run bfs22 --leg bfs --scale 22 --steps 64 --warmup 8
run bfs26 --leg bfs --scale 26 --steps 32 --warmup 8
find $out -name '*.csv' -size +6M -delete
find $out -name '*.db' -delete
