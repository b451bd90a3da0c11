# round-3 pass 1: GPU tests of the pruned / 64-bit-step / bool-done tree, A/B against the round-2 build, one bench line
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/r3_t1.log 2>&1; echo "pytest rc=$?" >> $O/r3_t1.log
tail -15 $O/r3_t1.log
timeout -k 10 300 python tools/exp/ab_fused.py --only shipped r2 --lib r2=tools/exp/libfpv_r2.so > $O/r3_ab1.log 2>&1; echo "ab rc=$?" >> $O/r3_ab1.log
cat $O/r3_ab1.log
timeout -k 10 300 python bench.py --steps 2000 --warmup 200 > $O/r3_bench1.json 2> $O/r3_bench1.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3_bench1.json').read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{d['value']/1e9:.2f} G/s  {r['avg_launch_us']:.2f} us  frac {r['frac']:.3f}  beyond {r['frac_beyond_mall']}", json.dumps(r["beyond_mall"]))
    print(json.dumps(d.get("cpu_baseline",{}))[:300])
except Exception as e: print("ERR", e); print(open('gpurun_out/r3_bench1.err').read()[-2000:])
PY
