# copies what tools/gpu/r3_final.sh left under gpurun_out/ into profiles/ (the tracked, judged copies)
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out; P=profiles
last() { tail -n 1 "$1" > "$2"; }
last $O/r3_bench_step.json              $P/r03_bench_n1_step.json
last $O/r3_bench_step_20.json           $P/r03_bench_n1_step_20steps.json
last $O/r3_bench_rollout.json           $P/r03_bench_n1_rollout.json
last $O/r3_bench_fp16.json              $P/r03_bench_n1_fp16.json
last $O/r3_bench_fp16_rollout.json      $P/r03_bench_n1_fp16_rollout.json
last $O/r3_bench_racerW.json            $P/r03_bench_n1_racerW.json
last $O/r3_bench_racerD.json            $P/r03_bench_n1_racerD.json
last $O/r3_bench_forcedist.json         $P/r03_bench_n1_forcedist.json
last $O/r3_bench_forcedist_20.json      $P/r03_bench_n1_forcedist_20steps.json
last $O/r3_bench_forcedist_rollout.json $P/r03_bench_n1_forcedist_rollout.json
cp $O/pmc_traffic.json $O/pmc_valu.json $O/r03_pmc_summary.md $O/r03_kernel_stats.csv $P/
cp $O/r3_valu_final.log   $P/r03_pmc_valu_counts.log
cp $O/r3_tests_final.log  $P/r03_gpu_tests.log
cp $O/r3_sweep_final.json $P/r03_sweep_variants.json
cp $O/r3_sweep_final.log  $P/r03_sweep_variants.log
cp $O/r3_sweep_4096.log   $P/r03_sweep_4096_drones.log
f=$(ls -t $O/prof_kt_variants/*/*_kernel_stats.csv | head -1); cp "$f" $P/r03_kernel_stats_variants.csv
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
h = bench.kernel_source_hash()
for f in ("profiles/pmc_traffic.json", "profiles/pmc_valu.json"):
    j = json.load(open(f))
    print(f, j["kernel_source_sha256_16"], "ok" if j["kernel_source_sha256_16"] == h else f"STALE (sources {h})")
PY
