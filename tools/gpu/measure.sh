# The ONE measurement pass of a round on one MI355X (run through gpurun AFTER the source freeze):
#     gpurun --timeout 1200 -- 'bash tools/gpu/measure.sh r06 counters' ; bash tools/gpu/collect.sh r06 counters
#     gpurun --timeout 1200 -- 'bash tools/gpu/measure.sh r06 bench'    ; bash tools/gpu/collect.sh r06 bench
#     gpurun --timeout 1150 -- 'bash tools/gpu/closed_loop.sh r06'      ; bash tools/gpu/collect.sh r06 loop
# (three calls: one gpurun call is at most 20 minutes; the bench stage quotes the traffic / VALU counts the counters stage put
# into profiles/ - collect in between)
# GPU tests, bench.py in every API and shape, kernel-trace stats, the separate FETCH_SIZE / WRITE_SIZE counter passes at the
# headline size, the counter groups beyond the Infinity Cache, the VALU instruction counts, the kernel sweep.  Everything
# lands under gpurun_out/<round>/; tools/gpu/collect.sh <round> copies the judged summaries into profiles/.
set -o pipefail
R=${1:-r06}
STAGE=${2:-all}
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
O=gpurun_out/$R; mkdir -p $O
run() { name=$1; shift; timeout -k 10 ${T:-300} "$@" > $O/$name.json 2> $O/$name.err; echo "$name rc=$?"; }
if [ "$STAGE" != bench ]; then
timeout -k 10 900 python -m pytest tests -m gpu -q > $O/gpu_tests.log 2>&1; echo "pytest rc=$?" >> $O/gpu_tests.log; tail -3 $O/gpu_tests.log
# wall-clock ratios: NOT part of the gate (marker gpu_timing); the measured ratios land in $O/timing_guards.json
FPV_TIMING_JSON=$O/timing_guards.json timeout -k 10 400 python -m pytest tests/test_gpu_timing.py -m gpu_timing -q > $O/gpu_timing.log 2>&1; echo "timing rc=$?" >> $O/gpu_timing.log; tail -2 $O/gpu_timing.log
python3 tools/gpu/device_props.py > $O/device_props.json 2> /dev/null; echo "props rc=$?"
timeout -k 10 200 python3 tools/xcd_map_probe.py --out $O/xcd_map.json > $O/xcd_map.log 2>&1; echo "xcd map rc=$?"
# counters first (they define traffic / VALU counts that the bench lines quote)
timeout -k 10 300 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/pmc_valu -- python3 tools/kernel_sweep.py --fp16 --racer --noise --extras --fused --rounds 1 --launches 64 --ring 32 > $O/pmc_valu.log 2>&1; echo "pmc valu rc=$?"
python3 tools/pmc_valu.py $O/pmc_valu --steps-per-launch 32 --round $R > $O/valu_counts.log 2>&1; tail -2 $O/valu_counts.log
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt -- python3 bench.py --steps 2000 --warmup 200 --no-cpu-baseline --no-beyond-mall > $O/prof_kt.log 2>&1; echo "kt rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_kt_variants -- python3 tools/kernel_sweep.py --fp16 --noise --extras --racer --fused --rounds 2 > $O/prof_kt_variants.log 2>&1; echo "kt variants rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -- python3 tools/pmc_probe.py > $O/pmc_fetch.log 2>&1; echo "fetch rc=$?"
timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -- python3 tools/pmc_probe.py > $O/pmc_write.log 2>&1; echo "write rc=$?"
python3 tools/pmc_summary.py --round $R --kt $O/prof_kt --fetch $O/pmc_fetch --write $O/pmc_write > $O/pmc_summary.log 2>&1; echo "summary rc=$?"
python3 tools/pmc_beyond_mall.py --print-groups | while read g ctrs; do
  timeout -k 10 200 rocprofv3 --kernel-trace --pmc $ctrs --output-format csv -d $O/pmcb/$g -- python3 tools/pmc_beyond_mall.py > $O/pmcb_$g.log 2>&1 < /dev/null; echo "pmcb $g rc=$?"
done
python3 tools/pmc_beyond_mall.py --summarise $O/pmcb --round $R > $O/pmcb_summary.log 2>&1; echo "pmcb summary rc=$?"
cp profiles/pmc_traffic.json profiles/pmc_valu.json profiles/${R}_pmc_summary.md profiles/${R}_kernel_stats.csv profiles/${R}_beyond_mall_counters.md $O/
fi
if [ "$STAGE" = counters ]; then exit 0; fi
# bench lines: the headline (default K / W), the driver's shape, every other API / storage / variant
T=500 run bench_step python bench.py
run bench_step_20 python bench.py --steps 20 --warmup 5
for i in 1 2 3; do run bench_p1_20_$i python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-beyond-mall --sustained-steps 0; run bench_p2_20_$i python bench.py --partitions 2 --steps 20 --warmup 5 --no-cpu-baseline --sustained-steps 0; done
run bench_partitions2 python bench.py --partitions 2 --no-cpu-baseline
run bench_rollout python bench.py --api rollout --no-cpu-baseline
run bench_fp16 python bench.py --fp16-state --no-cpu-baseline
run bench_fp16_rollout python bench.py --fp16-state --api rollout --no-cpu-baseline
run bench_racerW python bench.py --racer written --no-cpu-baseline --steps 5000
run bench_racerD python bench.py --racer omega_dt --no-cpu-baseline --steps 5000
run bench_forcedist python bench.py --force-dist --no-cpu-baseline --steps 5000
run bench_forcedist_20 python bench.py --force-dist --no-cpu-baseline --steps 20 --warmup 5
T=400 run bench_rehearsal_2ranks python bench.py --gpus 2 --rehearse-on-one-gpu --steps 2000 --warmup 100 --no-cpu-baseline --drones-per-gpu 524288
# the exact shape the driver launches at N = 2, both ranks on this one GPU over gloo: not a measurement, a rehearsal of the
# ABI-8 tree at world size 2 - supervisor, preflight, collective budget, the line's schema
T=400 run bench_rehearsal_2ranks_20 python bench.py --gpus 2 --rehearse-on-one-gpu --steps 20 --warmup 5
# where the state matrix lands decides the launch time beyond the Infinity Cache: the spread inside one process
timeout -k 10 300 python tools/beyond_placement.py combos 4 > $O/beyond_combos.log 2>&1; echo "combos rc=$?"
timeout -k 10 300 python tools/beyond_placement.py sizes 5 > $O/beyond_sizes.log 2>&1; echo "sizes rc=$?"
timeout -k 10 300 python examples/closed_loop_policy.py --hidden 0 > $O/closed_loop.log 2>&1 && timeout -k 10 300 python examples/closed_loop_policy.py --hidden 64 >> $O/closed_loop.log 2>&1; echo "closed loop rc=$?"
timeout -k 10 600 python tools/kernel_sweep.py --fp16 --noise --extras --racer --fused --ovr --aos --rounds 5 --out $O/sweep.json > $O/sweep.log 2>&1; echo "sweep rc=$?"
timeout -k 10 300 python tools/kernel_sweep.py --n 4096 --fused --graph --launches 256 --rounds 5 > $O/sweep_4096.log 2>&1; echo "sweep4096 rc=$?"
for f in $O/bench_*.json; do python3 - $f <<'PY'
import json,sys
try:
    d=json.loads(open(sys.argv[1]).read().strip().splitlines()[-1]); r=d["roofline"]
    b=r.get("beyond_mall") or {}; lf=r.get("launch_time_fit") or {}
    print(sys.argv[1].split("bench_")[1][:-5], f"{d['value']/1e9:.2f} G/s", f"{r['avg_launch_us']:.2f} us", f"host {r.get('host_enqueue_us', 0):.1f}", r["bound"], f"frac {r['frac']:.3f}",
          "beyond", b.get("frac"), "of copy", b.get("frac_of_copy_ceiling"), "fit", lf.get("floor_us"), lf.get("valid"), "traffic", r.get("traffic"), (d.get("cpu_baseline") or {}).get("value"))
except Exception as e: print(sys.argv[1], "ERR", e)
PY
done
