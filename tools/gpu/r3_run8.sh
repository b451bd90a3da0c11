set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q -x > $O/r3_t8.log 2>&1; echo "pytest rc=$?" >> $O/r3_t8.log; tail -6 $O/r3_t8.log
timeout -k 10 400 python tools/kernel_sweep.py --extras --noise --fused --rounds 5 > $O/r3_sweep8.log 2>&1; echo "sweep rc=$?"; cat $O/r3_sweep8.log
