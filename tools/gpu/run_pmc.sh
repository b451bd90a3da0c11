# kernel-trace stats of the bench command + the separate FETCH_SIZE / WRITE_SIZE counter passes (refreshes
# profiles/r02_kernel_stats.csv, r02_pmc_summary.md, pmc_traffic.json through tools/pmc_summary.py)
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
rm -rf $O/prof_kt $O/pmc_fetch $O/pmc_write
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-beyond-mall > $O/prof_kt.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_probe.py > $O/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/pmc_probe.py > $O/pmc_write.log 2>&1
echo "chain rc=$?"
