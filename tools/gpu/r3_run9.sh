# grouped-float4 state layout (4 group rows: 4 loads + 4 stores per lane instead of 14 + 14) vs the SoA rows, at 2^20 and 2^23 drones
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 300 python tools/exp/run_exp.py --n 1048576 --launches 200 --rounds 6 --pads 256 --variants 101,500,501,502,3 --check > $O/r3_exp_grp_2p20.log 2>&1; echo "rc=$?"; cat $O/r3_exp_grp_2p20.log
timeout -k 10 400 python tools/exp/run_exp.py --n 8388608 --ring 4 --launches 60 --rounds 5 --pads 256 --variants 101,500,501,502 --check > $O/r3_exp_grp_2p23.log 2>&1; echo "rc=$?"; cat $O/r3_exp_grp_2p23.log
