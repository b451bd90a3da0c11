# which runtime switches move the per-launch floor of a dependent kernel chain?  bench.py's own loop (5000 steps at 2^20 drones), one process per setting
set -o pipefail
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
run() { env "$@" timeout -k 5 120 python bench.py --steps 5000 --warmup 200 --no-cpu-baseline --no-beyond-mall --sustained-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('%-44s %8.3f us per step  %6.2f G env-steps/s' % (sys.argv[1], d['ms_per_step']*1e3, d['value']/1e9))" "$*"; }
run FPV_NONE=1
# AMD_OPT_FLUSH=0 (system-scope fences at every kernel boundary) was 24.26 against 22.41 us in the first try
# ROC_SYSTEM_SCOPE_SIGNAL=0 is NOT in the sweep: with it the host never sees the completion signals and torch.cuda.synchronize() hangs (first try)
run AMD_DIRECT_DISPATCH=0
run GPU_MAX_HW_QUEUES=1
run GPU_MAX_HW_QUEUES=8
run ROC_USE_FGS_KERNARG=0
run HIP_FORCE_DEV_KERNARG=1
run ROC_ACTIVE_WAIT_TIMEOUT=100
run DEBUG_CLR_KERNARG_HDP_FLUSH_WA=0
run FPV_NONE=2
