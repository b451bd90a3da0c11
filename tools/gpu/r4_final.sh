# round-4 measurement pass on one MI355X (run through gpurun): tests, bench in every API, kernel sweep,
# rocprofv3 kernel-trace stats, the separate FETCH_SIZE / WRITE_SIZE counter passes and the VALU instruction counts.
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/r4_tests_final.log 2>&1; echo "pytest rc=$?" >> $O/r4_tests_final.log; tail -4 $O/r4_tests_final.log
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_valu_f -- python3 tools/kernel_sweep.py --fp16 --racer --noise --extras --fused --rounds 1 --launches 64 --ring 32 > $O/pmc_valu_f.log 2>&1; echo "pmc valu rc=$?"
python3 tools/pmc_valu.py $O/pmc_valu_f --steps-per-launch 32 --round r04 > $O/r4_valu_final.log 2>&1; cp profiles/pmc_valu.json $O/pmc_valu.json; tail -3 $O/r4_valu_final.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-beyond-mall > $O/prof_kt.log 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_variants -- python3 tools/kernel_sweep.py --fp16 --noise --extras --racer --fused --rounds 2 > $O/prof_kt_variants.log 2>&1; echo "kt variants rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_probe.py > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/pmc_probe.py > $O/pmc_write.log 2>&1; echo "write rc=$?"
python3 tools/pmc_summary.py --round r04 --kt $O/prof_kt --fetch $O/pmc_fetch --write $O/pmc_write > $O/r4_pmc_summary.log 2>&1; echo "summary rc=$?"; cp profiles/pmc_traffic.json profiles/r04_pmc_summary.md profiles/r04_kernel_stats.csv $O/ 2>/dev/null
timeout -k 10 400 python bench.py > $O/r4_bench_step.json 2> $O/r4_bench_step.err; echo "bench step rc=$?"
timeout -k 10 300 python bench.py --steps 20 --warmup 5 > $O/r4_bench_step_20.json 2> $O/r4_bench_step_20.err; echo "bench 20 rc=$?"
timeout -k 10 300 python bench.py --api rollout --no-cpu-baseline > $O/r4_bench_rollout.json 2> $O/r4_bench_rollout.err; echo "bench rollout rc=$?"
timeout -k 10 300 python bench.py --fp16-state --no-cpu-baseline > $O/r4_bench_fp16.json 2> $O/r4_bench_fp16.err; echo "bench fp16 rc=$?"
timeout -k 10 300 python bench.py --fp16-state --api rollout --no-cpu-baseline > $O/r4_bench_fp16_rollout.json 2> $O/r4_bench_fp16_rollout.err; echo "bench fp16 rollout rc=$?"
timeout -k 10 300 python bench.py --racer written --no-cpu-baseline --steps 5000 > $O/r4_bench_racerW.json 2> $O/r4_bench_racerW.err; echo "racerW rc=$?"
timeout -k 10 300 python bench.py --racer omega_dt --no-cpu-baseline --steps 5000 > $O/r4_bench_racerD.json 2> $O/r4_bench_racerD.err; echo "racerD rc=$?"
timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --steps 5000 > $O/r4_bench_forcedist.json 2> $O/r4_bench_forcedist.err; echo "forcedist rc=$?"
timeout -k 10 300 python bench.py --force-dist --no-cpu-baseline --steps 20 --warmup 5 > $O/r4_bench_forcedist_20.json 2> $O/r4_bench_forcedist_20.err; echo "forcedist20 rc=$?"
timeout -k 10 300 python bench.py --force-dist --api rollout --no-cpu-baseline --steps 5000 > $O/r4_bench_forcedist_rollout.json 2> $O/r4_bench_forcedist_rollout.err; echo "forcedist rollout rc=$?"
timeout -k 10 300 python bench.py --partitions 2 --no-cpu-baseline > $O/r4_bench_partitions2.json 2> $O/r4_bench_partitions2.err; echo "partitions2 rc=$?"
timeout -k 10 400 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 2000 --warmup 100 --no-cpu-baseline --drones-per-gpu 524288 > $O/r4_bench_rehearsal_2ranks.json 2> $O/r4_bench_rehearsal_2ranks.err; echo "rehearsal rc=$?"
timeout -k 10 300 python tools/exp/split_streams.py > $O/r4_exp_split_streams.log 2>&1; echo "split streams rc=$?"
timeout -k 10 300 python examples/closed_loop_policy.py --hidden 0 > $O/r4_exp_closed_loop.log 2>&1; timeout -k 10 300 python examples/closed_loop_policy.py --hidden 64 >> $O/r4_exp_closed_loop.log 2>&1; echo "closed loop rc=$?"
timeout -k 10 600 python tools/kernel_sweep.py --fp16 --noise --extras --racer --fused --ovr --aos --rounds 5 --out $O/r4_sweep_final.json > $O/r4_sweep_final.log 2>&1; echo "sweep rc=$?"
timeout -k 10 300 python tools/kernel_sweep.py --n 4096 --fused --graph --launches 256 --rounds 5 > $O/r4_sweep_4096.log 2>&1; echo "sweep4096 rc=$?"
for f in step step_20 rollout fp16 fp16_rollout racerW racerD forcedist forcedist_20 forcedist_rollout partitions2 rehearsal_2ranks; do python - $O/r4_bench_$f.json <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    b=r.get("beyond_mall") or {}
    print(sys.argv[1].split("bench_")[1], f"{d['value']/1e9:.1f} G/s", f"{r['avg_launch_us']:.2f} us/launch", r["bound"], f"frac {r['frac']:.3f}", "beyond", r.get("frac_beyond_mall"), "of copy", b.get("frac_of_copy_ceiling"), "traffic", r.get("traffic"), "coll", (d.get("collective") or {}).get("world_seen"), (d.get("collective") or {}).get("library_version"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
cat $O/r4_sweep_final.log | tail -60; tail -8 $O/r4_sweep_4096.log
