#!/usr/bin/env python3
"""The reference's simulator loop (/root/reference/src/core/simulator.py:53-59, :83-91, :156) without
the rendering, for N drones at once on an MI355X:

    drone = Drone(params); targets / obstacles / ground; drone.reset(...)
    for i in range(time_steps):
        object_list = [*targets, *obstacles, ground]
        [target.update() for target in targets]
        action = np.array([-0.1, 0.0, 0.0, 0.0])
        drone.step(action, wind_velocity_vector, object_list)

Usage:  python examples/simulator_headless.py --drones 4096 --steps 2000
"""
import argparse
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from fpyv_amd import load_params  # noqa: E402
from fpyv_amd.components import Cylinder, Drone, Ground, Target  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--drones", type=int, default=4096)
    ap.add_argument("--steps", type=int, default=2000)
    ap.add_argument("--fps", type=float, default=60.0, help="simulator.fps of the reference params (dt = 1/fps)")
    ap.add_argument("--seed", type=int, default=0)
    a = ap.parse_args()
    rng = np.random.default_rng(a.seed)
    params = load_params(fps=a.fps)                      # same params.yaml schema as the reference

    # world, as generators.py builds it from params["simulator"] (targets on a circular path, 5 cylinders)
    # (the reference's own constructor calls: generators.py:22-25, :33-37, simulator.py:58)
    targets = [Target(np.array([0.0, 0.0, 3.0]) + 0.1 * rng.standard_normal(3), 1.0, 5,
                      {"radius": 25.0, "resolution": 5500})]
    obstacles = [Cylinder(np.array([0.0, 0.0, 0.0]) + np.array([10.0, 10.0, 0.0]) * rng.standard_normal(3),
                          abs(2.0 + 0.5 * rng.standard_normal()), abs(10.0 + 5.0 * rng.standard_normal()), 10, 25, random=True)
                 for _ in range(5)]
    ground = Ground(size=60, resolution=50, random=True)

    drone = Drone(params, num_envs=a.drones, device="cuda:0")
    spread = rng.uniform([-20, -20, 5], [20, 20, 15], (a.drones, 3)).astype(np.float32)
    drone.reset(position=spread, velocity=np.array(params.init_velocity), ypr=np.array(params.init_orientation_deg))
    wind_velocity_vector = np.array([0.0, 0.0, 0.0])
    action = torch.tensor([-0.1, 0.0, 0.0, 0.0], device="cuda:0").expand(a.drones, 4).contiguous()
    crashed = torch.zeros(a.drones, dtype=torch.bool, device="cuda:0")

    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for i in range(a.steps):
        object_list = [*targets, *obstacles, ground]
        [t.update() for t in targets]
        drone.step(action, wind_velocity_vector, object_list, return_imu=False)
        crashed |= drone.done
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    p = drone.position
    print(f"{a.drones} drones x {a.steps} steps (dt = {params.dt * 1e3:.2f} ms) in {dt:.3f} s "
          f"= {a.drones * a.steps / dt / 1e6:.1f} M env-steps/s (host loop with per-step world update)")
    print(f"crashed into ground/obstacles/target at some point: {int(crashed.sum())} of {a.drones}")
    print(f"mean position {p.mean(dim=0).cpu().numpy().round(3)}, mean speed "
          f"{float(drone.velocity.norm(dim=1).mean()):.3f} m/s")


if __name__ == "__main__":
    main()
