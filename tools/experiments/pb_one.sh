mkdir -p gpurun_out/pbone; rm -rf gpurun_out/pbone/*
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pbone -o t -- python tools/experiments/pb_one.py ${1:-26} ${2:-0} 3 > gpurun_out/pbone/out.txt 2> gpurun_out/pbone/err.txt
tail -1 gpurun_out/pbone/out.txt
f=$(find gpurun_out/pbone -name 't_kernel_trace.csv' | head -1)
python tools/experiments/pb_one.py --read $f | tee gpurun_out/pbone/last_search.txt | head -60
find gpurun_out/pbone -name '*.db' -delete; find gpurun_out/pbone -name '*trace.csv' -size +5M -delete
