# every counter group of tools/pmc_beyond_mall.py as its own rocprofv3 pass (run on the GPU box through gpurun)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/r5_pmcb; mkdir -p $O
python3 tools/pmc_beyond_mall.py --print-groups | while read g ctrs; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/$g -- python3 tools/pmc_beyond_mall.py > $O/$g.log 2>&1 < /dev/null; echo "$g rc=$?"
done
python3 tools/pmc_beyond_mall.py --summarise $O --round r05 > $O/summary.log 2>&1; echo "summary rc=$?"; cat $O/summary.log
