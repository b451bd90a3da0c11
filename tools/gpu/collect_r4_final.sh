# copies what tools/gpu/r4_final.sh left under gpurun_out/ into profiles/ (the tracked, judged copies)
set -e
cd "$(dirname "$0")/../.."
O=gpurun_out; P=profiles
last() { tail -n 1 "$1" > "$2"; }
last $O/r4_bench_step.json              $P/r04_bench_n1_step.json
last $O/r4_bench_step_20.json           $P/r04_bench_n1_step_20steps.json
last $O/r4_bench_rollout.json           $P/r04_bench_n1_rollout.json
last $O/r4_bench_fp16.json              $P/r04_bench_n1_fp16.json
last $O/r4_bench_fp16_rollout.json      $P/r04_bench_n1_fp16_rollout.json
last $O/r4_bench_racerW.json            $P/r04_bench_n1_racerW.json
last $O/r4_bench_racerD.json            $P/r04_bench_n1_racerD.json
last $O/r4_bench_forcedist.json         $P/r04_bench_n1_forcedist.json
last $O/r4_bench_forcedist_20.json      $P/r04_bench_n1_forcedist_20steps.json
last $O/r4_bench_forcedist_rollout.json $P/r04_bench_n1_forcedist_rollout.json
last $O/r4_bench_partitions2.json       $P/r04_bench_n1_step_partitions2.json
last $O/r4_bench_rehearsal_2ranks.json  $P/r04_bench_rehearsal_2ranks_on_one_gpu.json
cp $O/r4_exp_split_streams.log $P/r04_exp_split_streams.log
cp $O/r4_exp_closed_loop.log   $P/r04_exp_closed_loop_split_phase.log
cp $O/pmc_traffic.json $O/pmc_valu.json $O/r04_pmc_summary.md $O/r04_kernel_stats.csv $P/
cp $O/r4_valu_final.log   $P/r04_pmc_valu_counts.log
cp $O/r4_tests_final.log  $P/r04_gpu_tests.log
cp $O/r4_sweep_final.json $P/r04_sweep_variants.json
cp $O/r4_sweep_final.log  $P/r04_sweep_variants.log
cp $O/r4_sweep_4096.log   $P/r04_sweep_4096_drones.log
f=$(ls -t $O/prof_kt_variants/*/*_kernel_stats.csv | head -1); cp "$f" $P/r04_kernel_stats_variants.csv
python3 - <<'PY'
import json, sys
sys.path.insert(0, ".")
import bench
h = bench.kernel_source_hash()
for f in ("profiles/pmc_traffic.json", "profiles/pmc_valu.json"):
    j = json.load(open(f))
    print(f, j["kernel_source_sha256_16"], "ok" if j["kernel_source_sha256_16"] == h else f"STALE (sources {h})")
PY
