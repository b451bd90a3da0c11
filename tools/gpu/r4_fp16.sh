# round 4: the re-encoded fp16 state (15-bit-mantissa v, smallest-three q): GPU tests, VALU counts, timing
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "fp16 or step_n or widen or strided" > $O/r4_fp16_tests.log 2>&1; echo "pytest rc=$?"; tail -6 $O/r4_fp16_tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_valu_h -- python3 tools/kernel_sweep.py --fp16 --fused --rounds 1 --launches 64 --ring 32 > $O/pmc_valu_h.log 2>&1; echo "pmc valu rc=$?"
python3 tools/pmc_valu.py $O/pmc_valu_h --steps-per-launch 32 --round r04 > $O/r4_valu_fp16.log 2>&1; cat $O/r4_valu_fp16.log
timeout -k 10 400 python tools/kernel_sweep.py --fp16 --fused --rounds 7 > $O/r4_sweep_fp16.log 2>&1; echo "sweep rc=$?"; tail -8 $O/r4_sweep_fp16.log
