set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
timeout -k 10 600 python -m pytest tests -m gpu -q > gpurun_out/r2_t2.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r2_t2.log
timeout -k 10 300 python bench.py > gpurun_out/r2_bench_step.json 2> gpurun_out/r2_bench_step.err && \
timeout -k 10 300 python bench.py --api rollout --no-cpu-baseline > gpurun_out/r2_bench_rollout.json 2> gpurun_out/r2_bench_rollout.err && \
timeout -k 10 300 python bench.py --fp16-state --no-cpu-baseline > gpurun_out/r2_bench_fp16.json 2> gpurun_out/r2_bench_fp16.err && \
timeout -k 10 300 python bench.py --racer written --no-cpu-baseline --steps 5000 > gpurun_out/r2_bench_racerW.json 2> gpurun_out/r2_bench_racerW.err && \
timeout -k 10 300 python bench.py --racer omega_dt --no-cpu-baseline --steps 5000 > gpurun_out/r2_bench_racerD.json 2> gpurun_out/r2_bench_racerD.err && \
timeout -k 10 400 python tools/kernel_sweep.py --geom 1x128 1x256 --fp16 --noise --extras --racer --fused --rounds 5 --out gpurun_out/r2_sweep.json > gpurun_out/r2_sweep.log 2>&1
tail -3 gpurun_out/r2_t2.log; cat gpurun_out/r2_bench_step.json; cat gpurun_out/r2_bench_rollout.json; tail -40 gpurun_out/r2_sweep.log
