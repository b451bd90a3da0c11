# copies what tools/gpu/measure.sh <round> left under gpurun_out/<round>/ into profiles/ (the tracked, judged copies)
set -e
R=${1:-r06}
STAGE=${2:-all}
cd "$(dirname "$0")/../.."
O=gpurun_out/$R; P=profiles
last() { tail -n 1 "$1" > "$2"; }
if [ "$STAGE" = loop ]; then
python3 tools/closed_loop_summary.py $R > $P/${R}_closed_loop.md
cp $O/closed_loop_ab.json $P/${R}_closed_loop_ab.json
for t in 1048576_rot-1 1048576_rot0 8388608_rot-1 8388608_rot0 mlp; do cp $O/cl_kt_${t}_kernel_stats.csv $P/${R}_closed_loop_kernel_stats_${t}.csv; done
exit 0
fi
if [ "$STAGE" != bench ]; then
cp $O/pmc_traffic.json $O/pmc_valu.json $O/${R}_pmc_summary.md $O/${R}_kernel_stats.csv $O/${R}_beyond_mall_counters.md $P/
cp $O/valu_counts.log      $P/${R}_pmc_valu_counts.log
cp $O/gpu_tests.log        $P/${R}_gpu_tests.log
cp $O/gpu_timing.log       $P/${R}_gpu_timing.log
cp $O/timing_guards.json   $P/${R}_timing_guards.json
cp $O/device_props.json    $P/${R}_device_props.json
python3 - $O/xcd_map.json $P/${R}_xcd_map.json <<'PY'
import json, sys
d = json.load(open(sys.argv[1]))
for v in d.values():                      # 64 equal numbers per scenario say no more than their set
    v["distinct_shifts"] = sorted(set(v.pop("shifts")))
json.dump(d, open(sys.argv[2], "w"), indent=1)
PY
f=$(ls -t $O/prof_kt_variants/*/*_kernel_stats.csv | head -1); cp "$f" $P/${R}_kernel_stats_variants.csv
fi
if [ "$STAGE" = counters ]; then exit 0; fi
last $O/bench_step.json              $P/${R}_bench_n1_step.json
last $O/bench_step_20.json           $P/${R}_bench_n1_step_20steps.json
last $O/bench_partitions2.json       $P/${R}_bench_n1_step_partitions2.json
last $O/bench_rollout.json           $P/${R}_bench_n1_rollout.json
last $O/bench_fp16.json              $P/${R}_bench_n1_fp16.json
last $O/bench_fp16_rollout.json      $P/${R}_bench_n1_fp16_rollout.json
last $O/bench_racerW.json            $P/${R}_bench_n1_racerW.json
last $O/bench_racerD.json            $P/${R}_bench_n1_racerD.json
last $O/bench_forcedist.json         $P/${R}_bench_n1_forcedist.json
last $O/bench_forcedist_20.json      $P/${R}_bench_n1_forcedist_20steps.json
last $O/bench_rehearsal_2ranks.json     $P/${R}_bench_rehearsal_2ranks_on_one_gpu_2000steps.json
last $O/bench_rehearsal_2ranks_20.json  $P/${R}_bench_rehearsal_2ranks_on_one_gpu.json
python3 - $O $P/${R}_ab_partitions_20steps.json <<'PY'
import json, sys
out = {}
for k in ("p1", "p2"):
    rows = []
    for i in (1, 2, 3):
        d = json.loads(open(f"{sys.argv[1]}/bench_{k}_20_{i}.json").read().strip().splitlines()[-1]); r = d["roofline"]
        rows.append({"ms_per_step": d["ms_per_step"], "event_us_per_step": r["avg_launch_us"], "host_enqueue_us_per_step": r["host_enqueue_us"]})
    out["partitions=2" if k == "p2" else "single chain"] = rows
json.dump(out, open(sys.argv[2], "w"), indent=1)
PY
cp $O/closed_loop.log      $P/${R}_exp_closed_loop_split_phase.log
cp $O/sweep.json           $P/${R}_sweep_variants.json
cp $O/sweep.log            $P/${R}_sweep_variants.log
cp $O/sweep_4096.log       $P/${R}_sweep_4096_drones.log
grep -v amdgpu.ids $O/beyond_combos.log > $P/${R}_exp_beyond_combos_state_x_action_run2.log
grep -v amdgpu.ids $O/beyond_sizes.log  > $P/${R}_exp_beyond_sizes_run2.log
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
for f in ("profiles/pmc_traffic.json", "profiles/pmc_valu.json"):
    j = json.load(open(f))
    print(f, j.get("library_sha256_16"), j["kernel_source_sha256_16"], "ok" if bench.measurement_is_current(j) else f"STALE (library now {bench.library_hash()})")
PY
