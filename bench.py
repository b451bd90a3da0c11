#!/usr/bin/env python3
"""Headline benchmark: env-steps/s of the fused HIP stepper at 2^20 drones per GPU, dt = 1 ms.

    python bench.py --gpus 1 --steps 200 --warmup 20
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
        --master-port P bench.py --gpus N --steps K --warmup W

Workload = BASELINE.json configs[2]: 1 048 576 drones per GPU, fp32, EMA-smoothed Gaussian stick
noise (the profile of /root/reference/tests/noise_smooth_test.py:6-12) generated on the device
BEFORE the timed region into a ring of action batches, so every timed step reads a different
16.8 MB action batch from HBM.  A "step" is one pass of the hot path = one kernel launch through
the C ABI (fpv_step / fpv_rollout) that advances every drone of the shard by one Drone.step.
With N > 1 GPUs the drones are sharded contiguously (weak scaling: 2^20 per GPU) and each step's
bit-packed done mask is all-gathered over RCCL, asynchronously and double-buffered.

Prints ONE JSON line on rank 0 (see the driver contract) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time

REPO = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, REPO)

HBM_PEAK_GBS = 8000.0   # MI355X HBM3E peak, /opt/skills/guides/MI355X_MICROARCH.md "HBM3E peak BW"


def cpu_baseline(params, seconds_budget=12.0):
    """The float64 C oracle (a port of Drone.step; the reference itself is Python and cannot
    travel) timed on this host: all cores via OpenMP, on a bounded sample of the same workload."""
    import numpy as np
    from oracle import oracle
    threads = oracle.max_threads()
    n, chunk = 1 << 18, 32
    x = np.random.default_rng(1234).standard_normal((chunk, n, 4))
    acts = np.empty_like(x)
    s = np.zeros((n, 4))
    for t in range(chunk):                      # noise_smooth_test.py:11 recurrence
        s = 0.9 * s + 0.1 * x[t]
        acts[t] = s
    del x
    st = oracle.drone_initial_state(n, params.init_position, params.init_velocity, params.init_orientation_deg)
    oracle.drone_run(params, st, acts[:2], threads=threads)          # warm-up (thread pool, page faults)
    steps_done, t0 = 0, time.perf_counter()
    while True:
        oracle.drone_run(params, st, acts, threads=threads)
        steps_done += acts.shape[0]
        el = time.perf_counter() - t0
        if el > seconds_budget:
            break
    all_cores = n * steps_done / el
    st1 = oracle.drone_initial_state(n // 8, params.init_position, params.init_velocity, params.init_orientation_deg)
    t0 = time.perf_counter()
    oracle.drone_run(params, st1, acts[:, : n // 8].copy(), threads=1)
    one_core = (n // 8) * acts.shape[0] / (time.perf_counter() - t0)
    return {"value": all_cores, "unit": "env-steps/s", "cores": threads, "kind": "port",
            "sample": f"{n} drones x {steps_done} steps of EMA-noise sticks, float64 C restatement of Drone.step "
                      f"(oracle/fpv_oracle.c), OpenMP over drones; 1-thread rate {one_core:.3e} env-steps/s; "
                      f"reference's own Python Drone.step, timed in the build container only (it cannot travel): 2.7e3-4.0e3 env-steps/s on one core (profiles/r01_reference_python_timing.json)",
            "one_thread_value": one_core, "host_cpus": os.cpu_count()}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20000)
    ap.add_argument("--warmup", type=int, default=200)
    ap.add_argument("--preheat-s", type=float, default=0.5,
                    help="seconds of untimed launches on a SCRATCH batch before the W warm-up steps, so the "
                         "GPU has left its idle clocks (the timed region is only K x ~25 us)")
    ap.add_argument("--drones-per-gpu", type=int, default=1 << 20)
    ap.add_argument("--ring", type=int, default=32, help="distinct pre-generated action batches")
    ap.add_argument("--dpl", type=int, default=0, help="drones per lane (0 = library default)")
    ap.add_argument("--api", choices=["rollout", "step"], default="step",
                    help="rollout: K launches from one C call; step: one Python env.step() per launch")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--ceiling", type=float, default=100.0, help="auto-reset when |z| exceeds this (m)")
    ap.add_argument("--no-auto-reset", action="store_true")
    ap.add_argument("--fp16-state", action="store_true",
                    help="BASELINE configs[3]: v,q,rates,thrust stored as binary16 (89 B/env-step); not the headline")
    ap.add_argument("--no-gather", action="store_true", help="skip the done-mask all-gather (N > 1)")
    ap.add_argument("--gather-block", type=int, default=64,
                    help="steps of done masks bucketed into one all-gather (N > 1)")
    ap.add_argument("--gather-returns", action="store_true",
                    help="N > 1: also all-gather the per-drone last episode return (fp32) once per bucket; turns on "
                         "episode bookkeeping (+16 B per env-step), so it is off for the headline numbers")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal: run the RCCL process group + all-gather path even with one rank")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from fpyv_amd import load_params, sticks
    from fpyv_amd.dist import DoneGather
    from fpyv_amd.env import DroneBatch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but WORLD_SIZE={world}; launch with torch.distributed.run")
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the stepper has no CPU path")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    multi = world > 1 or args.force_dist
    if multi:
        if "MASTER_ADDR" not in os.environ:
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29517", RANK="0", WORLD_SIZE="1")
        # RCCL prints a version banner on STDOUT when the first communicator comes up; the driver reads
        # exactly one JSON line there, so route fd 1 to stderr until the communicator exists
        sys.stdout.flush()
        saved_fd = os.dup(1)
        os.dup2(2, 1)
        try:
            dist.init_process_group(backend="nccl", device_id=dev)
            warm = torch.zeros(1, device=dev)
            dist.all_reduce(warm)
            torch.cuda.synchronize()
        finally:
            sys.stdout.flush()
            os.dup2(saved_fd, 1)
            os.close(saved_fd)

    n = args.drones_per_gpu
    # dt = 1 ms; lanes that hit the ground or leave |z| <= ceiling are re-initialised in-kernel
    # (BASELINE configs[4] semantics; costs no extra bytes), so a long run stays a flight workload
    params = load_params(fps=1000, ceiling=args.ceiling)
    env = DroneBatch(params, n, device=dev, auto_reset=not args.no_auto_reset, with_accel=False,
                     with_done_bits=multi, fp16_state=args.fp16_state,
                     track_episodes=bool(multi and args.gather_returns))
    if args.dpl:
        env.set_tuning(args.dpl)
    env.reset()

    total = args.steps + args.warmup
    ring = max(1, min(args.ring, total))
    # seeded per rank so that every GPU of the job sees different sticks (global drone ids differ)
    actions = sticks.ema_noise_device(ring, n, dev, seed=1234 + rank)

    gather = None
    if multi and not args.no_gather:
        gather = DoneGather((env.done_bits.numel(),), torch.int64, dev, block=args.gather_block)
    returns_all, returns_work = None, None
    if gather is not None and args.gather_returns:
        returns_all = torch.zeros(dist.get_world_size() * n, dtype=torch.float32, device=dev)

    def run(k, t_base):
        """k steps = k launches.  Without the gather: whole ring spans go through fpv_rollout."""
        if gather is None and args.api == "rollout":
            t = t_base
            while t < t_base + k:
                r0 = t % ring
                span = min(ring - r0, t_base + k - t)
                env.rollout(actions[r0:r0 + span])
                t += span
        else:
            for t in range(t_base, t_base + k):
                if gather is not None:
                    env._buf.done_bits = gather.row_ptr(t)
                env.step(actions[t % ring], return_imu=False)
                if gather is not None:
                    gather.step_done(t)
                    if returns_all is not None and (t + 1) % args.gather_block == 0:
                        nonlocal returns_work
                        if returns_work is not None:
                            returns_work.wait()
                        returns_work = dist.all_gather_into_tensor(returns_all, env.last_return, async_op=True)

    def fence():
        torch.cuda.synchronize()
        if multi:
            dist.barrier()
            torch.cuda.synchronize()

    if args.preheat_s > 0:
        scratch = DroneBatch(params, n, device=dev, with_accel=False, fp16_state=args.fp16_state)
        scratch.reset()
        t_end = time.perf_counter() + args.preheat_s
        while time.perf_counter() < t_end:
            scratch.rollout(actions)
            torch.cuda.synchronize()
        del scratch
    run(args.warmup, 0)
    fence()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record()                      # torch's current stream == the stream the kernels are launched on
    run(args.steps, args.warmup)
    ev1.record()
    if gather is not None:
        gather.flush(args.warmup + args.steps - 1)
        gather.drain()
        if returns_work is not None:
            returns_work.wait()
    fence()
    elapsed = time.perf_counter() - t0
    dev_ms = ev0.elapsed_time(ev1)

    if multi:
        t = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())

    assert bool(torch.isfinite(env.state).all()), "non-finite state after the benchmark"

    if rank == 0:
        bytes_per_step = env.algorithmic_bytes()                             # 133 B fp32 / 89 B fp16 state
        kernel_s = dev_ms * 1e-3 / args.steps                                # avg launch-to-launch on the stream
        achieved = bytes_per_step * n / kernel_s / 1e9
        traffic, traffic_src = None, None
        tp = os.path.join(REPO, "profiles", "pmc_traffic.json")
        if os.path.isfile(tp) and not args.fp16_state:
            tj = json.load(open(tp))
            traffic, traffic_src = tj.get("hbm_bytes_per_launch"), tj.get("source")
        out = {
            "metric": "env-steps/sec at N=1.05M drones, dt=1ms; 1/2/4/8 GPU + CPU ref",
            "value": n * world * args.steps / elapsed,
            "unit": "env-steps/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": elapsed * 1e3 / args.steps,
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "state_storage": "f16(v,q,rates,thrust)+f32(p)" if args.fp16_state else "f32",
            "config": {"workload": ("configs[3]: fp16 state / fp32 integrator, " if args.fp16_state else "configs[2]: ")
                       + "1.05M drones/GPU, EMA-noise sticks (noise_smooth_test profile), fp32 math, dt=1ms, "
                       + ("no auto-reset" if args.no_auto_reset else f"in-kernel auto-reset on ground contact or |z|>{args.ceiling:g} m"),
                       "drones_per_gpu": n, "global_drones": n * world, "action_ring": ring, "api": args.api,
                       "parallelism": f"shard{world}" + ("+allgather(done_bits x" + str(args.gather_block) + " steps"
                                                              + (", last_return" if args.gather_returns else "") + ")" if gather is not None else ""),
                       "drones_per_lane": args.dpl or "default"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         "kernel": "fpv_drone_step_h_kernel" if args.fp16_state else "fpv_drone_step_kernel", "algorithmic_bytes_per_env_step": bytes_per_step,
                         "avg_launch_us": kernel_s * 1e6,
                         "note": "avg = HIP-event time over the timed region / launches (includes inter-launch gaps)"},
        }
        if not args.no_cpu_baseline and world == 1:
            out["cpu_baseline"] = cpu_baseline(params)
        print(json.dumps(out), flush=True)
    if multi:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
