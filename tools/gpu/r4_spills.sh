# round 4: after the SGPR-spill cleanup - every GPU test, then the variant sweep (single-step + k-step) for the kernels that changed
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r4_tests_spills.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r4_tests_spills.log
timeout -k 10 600 python tools/kernel_sweep.py --fp16 --noise --extras --racer --fused --ovr --rounds 5 > $O/r4_sweep_spills.log 2>&1; echo "sweep rc=$?"; cat $O/r4_sweep_spills.log
