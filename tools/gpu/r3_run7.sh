# beyond the MALL: do wider per-lane row accesses (2 / 4 drones per lane, float2 / float4 rows) help at 2^23 drones?
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 500 python tools/exp/run_exp.py --n 8388608 --ring 4 --launches 60 --rounds 5 --pads 256 --variants 101,100,131,102,111,113,130,112,121,123,126,125,127,300 --check > $O/r3_exp_wide_2p23.log 2>&1; echo "rc=$?"; cat $O/r3_exp_wide_2p23.log
timeout -k 10 300 python tools/exp/run_exp.py --n 1048576 --launches 200 --rounds 5 --pads 256 --variants 101,113,123,126 > $O/r3_exp_wide_2p20.log 2>&1; echo "rc=$?"; cat $O/r3_exp_wide_2p20.log
