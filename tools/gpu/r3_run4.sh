set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 400 python tools/exp/ab_fused.py --only shipped vconst r2 --lib r2=tools/exp/libfpv_r2.so --extra vconst=x --rounds 10 > $O/r3_ab4.log 2>&1; echo "ab rc=$?" >> $O/r3_ab4.log
cat $O/r3_ab4.log
