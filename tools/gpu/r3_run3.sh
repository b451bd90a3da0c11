# round-3 pass 3: fp16 packer (pkrtz stochastic rounding), VALU issue-rate probe
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r3_t3.log 2>&1; echo "pytest rc=$?" >> $O/r3_t3.log
tail -12 $O/r3_t3.log
timeout -k 10 120 tools/exp/issue/issue_probe > $O/r3_issue_probe.log 2>&1; echo "probe rc=$?"; cat $O/r3_issue_probe.log
timeout -k 10 400 python tools/kernel_sweep.py --fp16 --fused --rounds 5 > $O/r3_sweep3.log 2>&1; echo "sweep rc=$?"; cat $O/r3_sweep3.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_valu3 -- python3 tools/kernel_sweep.py --fp16 --fused --rounds 1 --launches 64 --ring 32 > $O/pmc_valu3.log 2>&1; echo "pmc rc=$?"
python3 tools/pmc_valu.py $O/pmc_valu3 --steps-per-launch 32 > $O/r3_valu3.log 2>&1; cat $O/r3_valu3.log
timeout -k 10 300 python bench.py --fp16-state --no-cpu-baseline --steps 5000 > $O/r3_bench3_fp16.json 2> $O/r3_bench3_fp16.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3_bench3_fp16.json').read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"fp16: {d['value']/1e9:.2f} G/s  {r['avg_launch_us']:.2f} us/launch frac {r['frac']:.3f} beyond {r['frac_beyond_mall']}")
except Exception as e: print("ERR", e); print(open('gpurun_out/r3_bench3_fp16.err').read()[-2000:])
PY
