# round-4 check of the rebuilt stick-noise generator: GPU tests, VALU counts (counter-only pass), timing sweep
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "noise or step_n or config2 or 2_to_the_32 or split_phase or shard" > $O/r4_noise_tests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r4_noise_tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_valu_n -- python3 tools/kernel_sweep.py --noise --fused --rounds 1 --launches 64 --ring 32 > $O/pmc_valu_n.log 2>&1; echo "pmc valu rc=$?"
python3 tools/pmc_valu.py $O/pmc_valu_n --steps-per-launch 32 --round r04 > $O/r4_valu_noise.log 2>&1; cat $O/r4_valu_noise.log
timeout -k 10 400 python tools/kernel_sweep.py --noise --fused --rounds 5 > $O/r4_sweep_noise.log 2>&1; echo "sweep rc=$?"; tail -20 $O/r4_sweep_noise.log
