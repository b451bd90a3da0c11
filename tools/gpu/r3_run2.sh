# round-3 pass 2: k-step kernels without SGPR spills (late argument views), object pass with lazy motor coordinates
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r3_t2.log 2>&1; echo "pytest rc=$?" >> $O/r3_t2.log
tail -12 $O/r3_t2.log
timeout -k 10 400 python tools/exp/ab_fused.py --only shipped r2 p1 unroll2 waves7 libsqrt --lib r2=tools/exp/libfpv_r2.so p1=tools/exp/libfpv_p1.so \
   --extra unroll2=x waves7=x libsqrt=x > $O/r3_ab2.log 2>&1; echo "ab rc=$?" >> $O/r3_ab2.log
cat $O/r3_ab2.log
timeout -k 10 600 python tools/kernel_sweep.py --fp16 --noise --extras --racer --fused --rounds 5 --out $O/r3_sweep2.json > $O/r3_sweep2.log 2>&1; echo "sweep rc=$?" >> $O/r3_sweep2.log
cat $O/r3_sweep2.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_valu2 -- python3 tools/kernel_sweep.py --fp16 --racer --noise --fused --rounds 1 --launches 64 --ring 32 > $O/pmc_valu2.log 2>&1; echo "pmc rc=$?"
python3 tools/pmc_valu.py $O/pmc_valu2 --steps-per-launch 32 > $O/r3_valu2.log 2>&1; cat $O/r3_valu2.log; cp profiles/pmc_valu.json $O/pmc_valu.json
timeout -k 10 300 python bench.py --api rollout --no-cpu-baseline > $O/r3_bench2_rollout.json 2> $O/r3_bench2_rollout.err; echo "bench rc=$?"
python - <<'PY'
import json
try:
    d=json.loads(open('gpurun_out/r3_bench2_rollout.json').read().strip().splitlines()[-1]); r=d["roofline"]
    print(f"{d['value']/1e9:.2f} G/s  {r['avg_launch_us']:.2f} us/launch bound {r['bound']} frac {r['frac']:.3f}", json.dumps(r.get("valu")), r.get("valu_unavailable"))
except Exception as e: print("ERR", e); print(open('gpurun_out/r3_bench2_rollout.err').read()[-2000:])
PY
