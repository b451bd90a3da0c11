# round-2 final measurement pass on one MI355X (run through gpurun): tests, bench in every API, kernel sweep,
# rocprofv3 kernel-trace stats and the separate FETCH_SIZE / WRITE_SIZE counter passes.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 600 python -m pytest tests -m gpu -q > $O/r2_t3.log 2>&1; echo "pytest rc=$?" >> $O/r2_t3.log
timeout -k 10 300 python bench.py > $O/r2_bench_step.json 2> $O/r2_bench_step.err && \
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/r2_bench_step_driver_shape.json 2> $O/r2_bench_step_driver_shape.err && \
timeout -k 10 300 python bench.py --api rollout --no-cpu-baseline > $O/r2_bench_rollout.json 2> $O/r2_bench_rollout.err && \
timeout -k 10 300 python bench.py --fp16-state --no-cpu-baseline > $O/r2_bench_fp16.json 2> $O/r2_bench_fp16.err && \
timeout -k 10 300 python bench.py --racer written --no-cpu-baseline --steps 5000 > $O/r2_bench_racerW.json 2> $O/r2_bench_racerW.err && \
timeout -k 10 300 python bench.py --racer omega_dt --no-cpu-baseline --steps 5000 > $O/r2_bench_racerD.json 2> $O/r2_bench_racerD.err && \
timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --steps 5000 > $O/r2_bench_forcedist.json 2> $O/r2_bench_forcedist.err && \
timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --steps 20 --warmup 5 > $O/r2_bench_forcedist_driver_shape.json 2> $O/r2_bench_forcedist_driver_shape.err && \
timeout -k 10 300 python bench.py --force-dist --api rollout --no-cpu-baseline --steps 5000 > $O/r2_bench_forcedist_rollout.json 2> $O/r2_bench_forcedist_rollout.err && \
timeout -k 10 500 python tools/kernel_sweep.py --geom 1x128 1x256 --fp16 --noise --extras --racer --fused --rounds 5 --out $O/r2_sweep.json > $O/r2_sweep.log 2>&1 && \
timeout -k 10 300 python tools/kernel_sweep.py --n 4096 --geom 1x128 --fused --graph --launches 256 --rounds 5 --out $O/r2_sweep_4096.json > $O/r2_sweep_4096.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-beyond-mall > $O/prof_kt.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_variants -- python3 tools/kernel_sweep.py --geom 1x128 --fp16 --noise --extras --racer --fused --rounds 2 > $O/prof_kt_variants.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_probe.py > $O/pmc_fetch.log 2>&1 && \
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/pmc_probe.py > $O/pmc_write.log 2>&1 && \
bash tools/gpu/run_valu.sh > $O/r2_valu.log 2>&1
echo "chain rc=$?"
tail -3 $O/r2_t3.log; for f in step step_driver_shape rollout fp16 racerW racerD forcedist forcedist_driver_shape forcedist_rollout; do python - $O/r2_bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    print(sys.argv[1].split("bench_")[1], f"{d['value']/1e9:.1f} G/s", f"{r['avg_launch_us']:.2f} us/launch", f"frac {r['frac']:.3f}", "beyond", r.get("frac_beyond_mall"), r.get("valu",{}).get("frac"), "traffic", r.get("traffic"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
cat $O/r2_sweep.log | tail -50; cat $O/r2_sweep_4096.log | tail -6; tail -8 $O/r2_valu.log
