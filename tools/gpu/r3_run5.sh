set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 120 rocprofv3 -L > $O/r3_counters_avail.txt 2>&1; echo "list rc=$?"
grep -o "Name:\s*SQ_[A-Z0-9_]*" $O/r3_counters_avail.txt | sort -u | tr '\n' ' ' | head -c 6000; echo
