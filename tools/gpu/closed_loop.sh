# Closed-loop evidence: the rotation of the traversal inside action -> step -> action, and which kernels the
# policy half is.  gpurun --timeout 1100 -- 'bash tools/gpu/closed_loop.sh r06'
set -o pipefail
R=${1:-r06}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
python3 tools/gpu/device_props.py > $O/device_props.json 2> $O/device_props.err; echo "props rc=$?"
timeout -k 10 500 python3 tools/closed_loop_ab.py --out $O/closed_loop_ab.json > $O/closed_loop_ab.log 2>&1; echo "ab rc=$?"; tail -30 $O/closed_loop_ab.log
for n in 1048576 8388608; do for rot in -1 0; do
  steps=500; [ $n = 8388608 ] && steps=150
  timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl_kt_${n}_rot${rot} -- python3 examples/closed_loop_policy.py --drones $n --steps $steps --partitions 1 --rotation $rot > $O/cl_kt_${n}_rot${rot}.log 2>&1; echo "kt $n rot $rot rc=$?"
done; done
timeout -k 10 240 rocprofv3 --kernel-trace --stats --output-format csv -d $O/cl_kt_mlp -- python3 examples/closed_loop_policy.py --drones 1048576 --steps 300 --partitions 1 --hidden 64 > $O/cl_kt_mlp.log 2>&1; echo "kt mlp rc=$?"
for d in $O/cl_kt_*/; do f=$(ls -t $d*/*_kernel_stats.csv 2>/dev/null | head -1); [ -n "$f" ] && cp "$f" $O/$(basename $d)_kernel_stats.csv; done
ls $O
