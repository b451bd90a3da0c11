#!/usr/bin/env python3
"""The rotation of the traversal inside the loop the reference actually runs - action -> step -> action
(/root/reference/src/core/simulator.py:83-156) - and what the policy half of that loop costs on the stepper's layout.

One process, the same buffers, interleaved rounds (A B A B ...), HIP events on the stream the loop runs on:

  1. closed loop  tanh(W[4,13] @ obs[13,N]) -> fpv_step   with fpv_set_rotation(-1) (automatic) and (0) (plain order)
  2. the step-only chain of the same handle, both settings (the number every rotation claim so far was made on)
  3. the policy alone on (a) the live state view obs = state[:13, :n] (row stride ld = fpv_recommended_ld(n)),
     (b) a contiguous [13, n] copy, (c) the same with the product written into a preallocated [4, n] buffer,
     (d) the transposed formulation obs^T[N,13] @ W^T[13,4] -> [N,4] rows (what fpv_step reads with action_ld = 0)
     - is the padded row stride, or the SoA output, what the GEMM pays for?

    python tools/closed_loop_ab.py --drones 1048576 8388608 --out gpurun_out/r06/closed_loop_ab.json

Per-kernel attribution of the same loop comes from rocprofv3 --kernel-trace --stats on examples/closed_loop_policy.py
(--rotation -1 / 0); tools/closed_loop_summary.py puts both into profiles/r06_closed_loop.md."""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import load_params  # noqa: E402
from fpyv_amd.env import DroneBatch  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--drones", type=int, nargs="+", default=[1 << 20, 1 << 23])
ap.add_argument("--rounds", type=int, default=5)
ap.add_argument("--steps", type=int, default=0, help="closed-loop steps per round (0: 600 at <= 2^21 drones, 150 beyond)")
ap.add_argument("--out", default="")
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.cuda.set_stream(torch.cuda.Stream(device=dev))
torch.manual_seed(0)
W = torch.randn(4, 13, device=dev) * 0.02
Wt = W.t().contiguous()
ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)


def timed(fn, k):
    torch.cuda.synchronize()
    ev0.record()
    for _ in range(k):
        fn()
    ev1.record()
    torch.cuda.synchronize()
    return ev0.elapsed_time(ev1) * 1e3 / k


report = {"policy": "tanh(W[4,13] @ obs[13,N])", "rounds": a.rounds, "sizes": []}
for n in a.drones:
    steps = a.steps or (600 if n <= (1 << 21) else 150)
    env = DroneBatch(load_params(fps=1000, ceiling=100.0), n, device=dev, auto_reset=True)
    env.reset()
    obs = env.state[:13, :n]
    sticks_rows = torch.zeros((n, 4), device=dev)
    auto = None

    def closed():
        env.step(torch.tanh(W @ obs), return_imu=False)

    def step_only():
        env.step(sticks_rows, return_imu=False)

    res = {"closed_auto": [], "closed_plain": [], "step_auto": [], "step_plain": []}
    timed(closed, 50)
    for _ in range(a.rounds):
        for name, rot, fn in (("closed_auto", -1, closed), ("closed_plain", 0, closed), ("step_auto", -1, step_only), ("step_plain", 0, step_only)):
            env.set_rotation(rot)
            timed(fn, 30)                                        # settle the caches into this order
            res[name].append(timed(fn, steps))
            if rot == -1:
                auto = env.rotation
    env.set_rotation(-1)
    assert bool(torch.isfinite(env.state).all())

    # the policy alone
    obs_c = obs.contiguous()
    out_soa = torch.empty((4, n), device=dev)
    out_rows = torch.empty((n, 4), device=dev)
    pol = {
        "gemm_on_state_view": lambda: W @ obs,
        "gemm_on_contiguous_copy": lambda: W @ obs_c,
        "gemm_on_state_view_out_preallocated": lambda: torch.mm(W, obs, out=out_soa),
        "gemm_transposed_rows_out": lambda: torch.mm(obs.t(), Wt, out=out_rows),
        "gemm_transposed_rows_out_contiguous": lambda: torch.mm(obs_c.t(), Wt, out=out_rows),
        "tanh_4xN": lambda: torch.tanh(out_soa),
        "tanh_inplace_4xN": lambda: torch.tanh_(out_soa),
        "policy_on_state_view": lambda: torch.tanh(W @ obs),
        "policy_on_contiguous_copy": lambda: torch.tanh(W @ obs_c),
    }
    pres = {k: [] for k in pol}
    for _ in range(a.rounds):
        for k, fn in pol.items():
            timed(fn, 10)
            pres[k].append(timed(fn, 100))
    size = {"drones": n, "ld": env.ld, "state_view_row_stride_bytes": 4 * env.ld, "rotation_auto_drones": auto, "steps_per_round": steps,
            "us_per_step": {k: {"median": statistics.median(v), "min": min(v), "max": max(v)} for k, v in res.items()},
            "policy_us_per_call": {k: {"median": statistics.median(v), "min": min(v), "max": max(v)} for k, v in pres.items()},
            "policy_algorithmic_bytes": {"gemm": 13 * 4 * n + 16 * n, "tanh": 32 * n}}
    m = size["us_per_step"]
    size["closed_loop_rotation_gain"] = m["closed_plain"]["median"] / m["closed_auto"]["median"]
    size["step_only_rotation_gain"] = m["step_plain"]["median"] / m["step_auto"]["median"]
    report["sizes"].append(size)
    print(f"{n} drones (ld {env.ld}, rotation {auto}): closed loop {m['closed_auto']['median']:.1f} us auto / {m['closed_plain']['median']:.1f} us plain; "
          f"step only {m['step_auto']['median']:.2f} / {m['step_plain']['median']:.2f}", flush=True)
    for k, v in size["policy_us_per_call"].items():
        print(f"    {k:42s} {v['median']:8.2f} us  ({v['min']:.2f} - {v['max']:.2f})", flush=True)
    env.close()
    del env, obs, obs_c, out_soa, out_rows, sticks_rows
    torch.cuda.empty_cache()
if a.out:
    os.makedirs(os.path.dirname(a.out) or ".", exist_ok=True)
    with open(a.out, "w") as f:
        json.dump(report, f, indent=1)
