# round 4: after trimming the physics' instruction mix (fpv_rot 27 -> 20, low-pass gains folded): tests, VALU counts, timing
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r4_mix_tests.log 2>&1; echo "pytest rc=$?"; tail -4 $O/r4_mix_tests.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_valu_m -- python3 tools/kernel_sweep.py --fp16 --noise --racer --fused --rounds 1 --launches 64 --ring 32 > $O/pmc_valu_m.log 2>&1; echo "pmc valu rc=$?"
python3 tools/pmc_valu.py $O/pmc_valu_m --steps-per-launch 32 --round r04 > $O/r4_valu_mix.log 2>&1; cat $O/r4_valu_mix.log
timeout -k 10 400 python tools/kernel_sweep.py --fp16 --noise --fused --rounds 7 > $O/r4_sweep_mix.log 2>&1; echo "sweep rc=$?"; tail -10 $O/r4_sweep_mix.log
