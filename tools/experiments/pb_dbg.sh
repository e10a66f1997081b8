mkdir -p gpurun_out/pbone
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for d in 0 1 2 4 7; do
rm -rf gpurun_out/pbone/*
FGPU_PB_DBG=$d rocprofv3 --kernel-trace --output-format csv -d gpurun_out/pbone -o t -- python tools/experiments/pb_one.py 26 0 2 > gpurun_out/pbone/out.txt 2> gpurun_out/pbone/err.txt
f=$(find gpurun_out/pbone -name 't_kernel_trace.csv' | head -1)
echo "dbg=$d: $(python tools/experiments/pb_one.py --read $f | grep bfs_pb_apply | sort -k3 -n -r | head -1)"
done
rm -rf gpurun_out/pbone
