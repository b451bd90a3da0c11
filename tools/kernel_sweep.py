#!/usr/bin/env python3
"""Interleaved A/B timing of the kernel variants and host paths in ONE process (cdna_hip_programming.md 5.4 rule 24).
Prints median / min microseconds per step and the algorithmic-bytes bandwidth it implies.  (Rounds 1-2 also swept
launch geometries here - 2 / 4 drones per lane, 256-thread workgroups; they lost everywhere and are no longer built.)"""
import argparse
import json
import os
import statistics
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

from fpyv_amd import _lib, load_params, sticks  # noqa: E402
from fpyv_amd.env import DroneBatch, RacerBatch  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--n", type=int, default=1 << 20)
    ap.add_argument("--launches", type=int, default=200)
    ap.add_argument("--rounds", type=int, default=7)
    ap.add_argument("--ring", type=int, default=32)
    ap.add_argument("--fp16", action="store_true", help="also time the fp16-storage kernel (config 4, 89 B/env-step)")
    ap.add_argument("--aos", action="store_true", help="also time the step with the [n,16] AoS observation head")
    ap.add_argument("--noise", action="store_true", help="also time pure in-kernel noise sticks (no action read)")
    ap.add_argument("--extras", action="store_true", help="also time the Kahan-row and 4-object collision variants")
    ap.add_argument("--graph", action="store_true", help="also time the hipGraph-replayed rollout")
    ap.add_argument("--fused", action="store_true", help="also time the k-step kernel (fpv_step_n), k = ring span per launch")
    ap.add_argument("--racer", action="store_true", help="also time the Racer kernels (as written / omega*dt / components.PID)")
    ap.add_argument("--ovr", action="store_true", help="also time the guidance-override step (rotation_matrix= / thrust_force=)")
    ap.add_argument("--out", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda:0")
    # the bench's workload: auto-reset on ground contact or |z| > 100 m.  Without it the population does not stay a
    # flight workload - after a few seconds of EMA-noise sticks the drones have tumbled, half of them have gone through the
    # ground plane and keep falling, and the object pass then measures "half the drones below the ground" instead of
    # "nobody near" (rounds 1-2 swept without it: the bimodal object-list numbers of those logs)
    p = load_params(fps=1000, ceiling=100.0)
    acts = sticks.ema_noise_device(a.ring, a.n, dev)
    envs = {}
    # ONE state buffer for every geometry: timings depend on buffer placement
    shared = DroneBatch(p, a.n, device=dev, with_accel=False, auto_reset=True)
    shared.reset()
    a.geom = ["f32"]
    envs["f32"] = shared
    if a.fp16:
        e16 = DroneBatch(p, a.n, device=dev, with_accel=False, fp16_state=True, auto_reset=True)
        e16.reset()
        envs["h"] = e16                                         # fp16 storage (half2 pair rows)
        a.geom = list(a.geom) + ["h"]
    if a.aos:
        ea = DroneBatch(p, a.n, device=dev, with_accel=False, with_obs_aos=True, auto_reset=True)
        ea.reset()
        envs["aos"] = ea
        a.geom = list(a.geom) + ["aos"]
    if a.noise:
        en = DroneBatch(p, a.n, device=dev, with_accel=False, stick_noise=True, noise_seed=1, auto_reset=True)
        en.reset()
        envs["noise"] = en
        a.geom = list(a.geom) + ["noise"]
    if a.extras:
        from fpyv_amd.objects import Cylinder, Ground, Target
        ek = DroneBatch(p, a.n, device=dev, with_accel=False, kahan_position=True, auto_reset=True)
        ek.reset()
        envs["kahan"] = ek
        eo = DroneBatch(p, a.n, device=dev, with_accel=False, auto_reset=True)
        eo.reset()
        world = [Target([0, -6, 3], 0.8), Cylinder([3, 0, 0], 1.0, 5.0), Cylinder([-2, 2.5, 0], 0.6, 1.5), Ground()]
        envs["obj"] = eo
        # the same kind of list a kilometre away and without a ground plane: four objects, nobody ever near - what the
        # object pass costs a population that is not interacting with its world
        ef = DroneBatch(p, a.n, device=dev, with_accel=False, auto_reset=True)
        ef.reset()
        far = [Target([1000, 994, 3], 0.8), Cylinder([1003, 1000, 0], 1.0, 5.0), Cylinder([998, 1002.5, 0], 0.6, 1.5), Cylinder([1010, 990, 0], 2.0, 30.0)]
        envs["objfar"] = ef
        a.geom = list(a.geom) + ["kahan", "obj", "objfar"]
    if a.racer:
        import numpy as np
        pid = np.array([[0.004, 0.02, 1e-6], [0.003, 0.01, 2e-6], [0.002, 0.005, 0.0]])
        for tag, kw in (("racerW", dict(racer_omega_dt=False)), ("racerD", dict(racer_omega_dt=True)),
                        ("racerWC", dict(racer_omega_dt=False, racer_pid_variant=1, racer_pid=-pid, pid_integral_clip=0.05,
                                         pid_min_output=-0.004, pid_max_output=0.006, pid_derivative_transition_rate=0.3))):
            rp = p.replace(**dict(dict(mode=1, racer_pid=pid, ceiling=50.0), **kw))
            er = RacerBatch(rp, a.n, device=dev, auto_reset=True)
            er.reset()
            envs[tag] = er
            a.geom = list(a.geom) + [tag]
    if a.ovr:
        eg = DroneBatch(p, a.n, device=dev, with_accel=False, auto_reset=True)
        eg.reset()
        ang = torch.rand(a.n, device=dev) * 0.3
        zero, one = torch.zeros_like(ang), torch.ones_like(ang)
        rot_over = torch.stack([ang.cos(), zero, ang.sin(), zero, one, zero, -ang.sin(), zero, ang.cos()], dim=1).contiguous()   # Ry
        thrust_over = torch.full((a.n,), 7.4, device=dev)
        eg._buf.rotation_override, eg._buf.thrust_override = rot_over.data_ptr(), thrust_over.data_ptr()
        envs["ovr"] = eg
        a.geom = list(a.geom) + ["ovr"]
    variants = [(g, api) for g in a.geom for api in ("rollout", "step") if not (g.startswith("ovr") and api == "rollout")]
    if a.graph:
        variants += [(g, "graph") for g in a.geom if g == "f32"]
    if a.fused:
        variants += [(g, "fused") for g in a.geom if g not in ("aos", "ovr")]
    times = {v: [] for v in variants}
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for r in range(a.rounds + 1):
        for v in variants:
            d, api = v
            e = envs[d]
            torch.cuda.synchronize()
            ev0.record()
            done = 0
            while done < a.launches:
                span = min(a.ring, a.launches - done)
                if d.startswith("noise"):
                    if api in ("rollout", "fused"):
                        e.rollout(None, steps=span, fused=(api == "fused"))
                    else:
                        for t in range(span):
                            e.step(None, return_imu=False)
                elif d.startswith("obj"):
                    w = far if d == "objfar" else world
                    if api in ("rollout", "fused"):
                        e.set_objects(w)
                        e.rollout(acts[:span], fused=(api == "fused"))
                    else:
                        e.set_objects(w)                      # bound once per span: the per-step Python conversion of the
                        for t in range(span):                  # list (30 us) would hide a 24 us kernel
                            e._step_raw(acts[t])
                elif d.startswith("ovr"):
                    for t in range(span):
                        e._step_raw(acts[t])                   # the override pointers stay bound in e._buf
                elif api == "graph":
                    e.rollout(acts[:span], graph=True)
                elif api in ("rollout", "fused"):
                    e.rollout(acts[:span], fused=(api == "fused"))
                else:
                    for t in range(span):
                        e.step(acts[t], return_imu=False)
                done += span
            ev1.record()
            torch.cuda.synchronize()
            if r:   # round 0 = warm-up;  "launches" counts env steps: a fused launch advances `span` of them
                times[v].append(ev0.elapsed_time(ev1) * 1e3 / a.launches)
    res = []
    for v in variants:
        B = envs[v[0]].algorithmic_bytes() + (64 if v[0].startswith("aos") else 0) + (16 if v[0].startswith("noise") else 0) + (48 if v[0].startswith("kahan") else 0) + (40 if v[0].startswith("ovr") else 0)
        if v[1] == "fused":        # state traffic amortised over the ring span; reward/done only after the last step
            B = 16 + (B - 16) / min(a.ring, a.launches) if not v[0].startswith("noise") else (B - 16) / min(a.ring, a.launches)
        med, mn = statistics.median(times[v]), min(times[v])
        res.append({"geom": v[0], "api": v[1], "median_us": med, "min_us": mn,
                    "GBps_alg_median": B * a.n / med / 1e3, "env_steps_per_s_median": a.n / med * 1e6})
        print(f"geom={v[0]} api={v[1]:8s} median {med:8.2f} us/step  min {mn:8.2f}  "
              f"{B * a.n / med / 1e3:8.1f} GB/s(alg)  {a.n / med:8.1f} M env-steps/s", flush=True)
    if a.out:
        json.dump({"n": a.n, "launches": a.launches, "rounds": a.rounds, "results": res}, open(a.out, "w"), indent=1)


if __name__ == "__main__":
    main()
